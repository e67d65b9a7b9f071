"""Small helpers shared by the hot-path facade (reference: poreplex/utils.py)."""

__all__ = ['union_intervals']


def union_intervals(iset):
    """Merge overlapping / touching [begin, end] intervals (utils.py:28-39);
    used by the pseudo-fusion filter (signal_analyzer.py:422-424)."""
    merged = []
    for begin, end in sorted(iset):
        if merged and merged[-1][1] >= begin:
            if merged[-1][1] < end:
                merged[-1][1] = end
            continue
        merged.append([begin, end])
    return merged
