"""Per-process model/context cache (reference: poreplex/worker_persistence.py).

The reference keeps the pomegranate HMMs, the two Keras models and the poly(A)
analyzer alive for the life of a worker process by hiding them on a fake module
in sys.modules (:37-38,85-88).  Here the thing that must live that long is the
GPU context (weights + HMM tables resident in HBM, stream, arenas): one
``NativeContext`` per worker process and GPU.
"""
import os
import sys
import threading
import types

from . import native

__all__ = ['WorkerPersistenceStorage']

_INIT_LOCK = threading.Lock()


class WorkerPersistenceStorage:

    STORAGE_NAME = '__poreplex_amd_persistence'
    VARIABLES = ['segmodel', 'unsplitmodel', 'kmermodel', 'kmersize', 'loader',
                 'demuxer', 'albacore', 'polyaanalyzer', 'ctx']

    def __init__(self, config):
        self.config = config

    def retrieve_objects(self, target):
        with _INIT_LOCK:              # worker calls may arrive on several threads at once
            if self.STORAGE_NAME not in sys.modules:
                storage = self.init_persistence_objects(self.config)
            else:
                storage = sys.modules[self.STORAGE_NAME].storage
        for varname in self.VARIABLES:
            if varname in storage:
                setattr(target, varname, storage[varname])
        for varname in ('loader', 'demuxer'):       # worker_persistence.py:56-58
            if varname in storage:
                storage[varname].clear()

    def init_persistence_objects(self, config):
        from .barcoding import BarcodeDemultiplexer
        from .polya import PolyASignalAnalyzer
        from .signal_loader import SignalLoader
        from .segmentation import SegmentationModel

        device_id = int(config.get('device_id', os.environ.get('LOCAL_RANK', 0)))
        ctx = native.NativeContext(config, device_id=device_id)
        storage = {
            'ctx': ctx,
            'segmodel': SegmentationModel(ctx, 0),
            'unsplitmodel': SegmentationModel(ctx, 1),
            'kmermodel': None,
            'kmersize': 5,        # the kmer_models submodule is not shipped
        }
        if config['barcoding']:
            storage['demuxer'] = BarcodeDemultiplexer(config['demultiplexing'],
                                                      config['barcoding_quality_filter'], ctx)
        if config['measure_polya']:
            storage['polyaanalyzer'] = PolyASignalAnalyzer(config['polya_dwell'], ctx)
        if config['albacore_onthefly']:
            raise NotImplementedError('on-the-fly albacore basecalling is out of scope')
        storage['loader'] = SignalLoader(config['signal_processing'], config['inputdir'], ctx,
                                         config.get('read_bundle'))
        mod = types.ModuleType(self.STORAGE_NAME)
        mod.storage = storage
        sys.modules[self.STORAGE_NAME] = mod
        return storage

    @classmethod
    def reset(cls):
        """Drop the cached context (tests / device switch)."""
        mod = sys.modules.pop(cls.STORAGE_NAME, None)
        if mod is not None and 'ctx' in mod.storage:
            if 'loader' in mod.storage:
                mod.storage['loader'].unpin_bundle()
            mod.storage['ctx'].close()


def warm_up(config):
    """Build this worker process's persistent objects NOW: what the first process_batch call of a fresh worker would do
    (context on the GPU, models, the read bundle).  For a pool initializer --
    `ProcessPoolExecutor(workers, initializer=warm_up, initargs=(config,))` -- so that no batch pays for it."""
    class _Sink:
        pass
    WorkerPersistenceStorage(config).retrieve_objects(_Sink())
