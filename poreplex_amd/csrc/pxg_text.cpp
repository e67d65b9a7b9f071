// pxg_text.cpp -- libpxghost.so, host only: the rows of sequencing_summary.txt (io.py:120-184,
// SequencingSummaryWriter) formatted from columns.  After the kernels went from ~70 to ~15 ms
// per 10 000-read batch the session's main thread was the slowest stage of the end-to-end
// pipeline, and a third of it was Python building this text field by field (1.05 us per
// read).  The text must stay what Python prints: str(int), repr(float) -- the shortest string
// that round-trips, in repr's fixed / exponent layout -- round(x, 3) and format(x, '.4f').
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../include/pxg.h"

namespace {

struct Out {
    char* p;
    char* end;
    bool ok = true;
    void put(char c)
    {
        if (p < end) *p++ = c; else ok = false;
    }
    void put(const char* s)
    {
        put(s, strlen(s));
    }
    void put(const char* s, size_t n)
    {
        if ((size_t)(end - p) >= n) { memcpy(p, s, n); p += n; } else ok = false;
    }
    void put_int(int64_t v)
    {
        char buf[24];
        auto r = std::to_chars(buf, buf + sizeof buf, v);
        put(buf, (size_t)(r.ptr - buf));
    }
};

// one element of a NumPy '<U' array (UCS-4, fixed width, NUL padded); false if not ASCII
bool put_ucs4(Out& o, const uint32_t* s, int64_t width)
{
    if (o.end - o.p < width) {
        o.ok = false;
        return true;
    }
    char* p = o.p;
    uint32_t seen = 0;
    int64_t i = 0;
    for (; i < width && s[i]; i++) {
        seen |= s[i];
        p[i] = (char)s[i];
    }
    o.p = p + i;
    return seen < 128;
}

// repr(float): shortest round-trip digits (std::to_chars), laid out the way CPython's
// format_float_short does for 'r': exponent form iff decpt <= -4 or decpt > 16
void put_repr(Out& o, double x)
{
    if (std::isnan(x)) { o.put("nan"); return; }
    if (std::isinf(x)) { o.put(x < 0 ? "-inf" : "inf"); return; }
    if (std::signbit(x)) { o.put('-'); x = -x; }
    if (x == 0.0) { o.put("0.0"); return; }
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);   // d[.ddd]e[+-]XX
    char digits[24];
    int nd = 0;
    const char* q = buf;
    for (; q < r.ptr && *q != 'e'; q++)
        if (*q != '.') digits[nd++] = *q;
    int ex = 0;
    {
        const char* e = q + 1;
        const bool neg = *e == '-';
        if (*e == '-' || *e == '+') e++;
        for (; e < r.ptr; e++) ex = ex * 10 + (*e - '0');
        if (neg) ex = -ex;
    }
    const int decpt = ex + 1;
    if (decpt <= -4 || decpt > 16) {
        o.put(digits[0]);
        if (nd > 1) { o.put('.'); o.put(digits + 1, (size_t)nd - 1); }
        o.put('e');
        int e10 = decpt - 1;
        o.put(e10 < 0 ? '-' : '+');
        if (e10 < 0) e10 = -e10;
        if (e10 < 10) o.put('0');
        o.put_int(e10);
    } else if (decpt <= 0) {
        o.put("0.");
        for (int i = 0; i < -decpt; i++) o.put('0');
        o.put(digits, (size_t)nd);
    } else if (decpt >= nd) {
        o.put(digits, (size_t)nd);
        for (int i = nd; i < decpt; i++) o.put('0');
        o.put(".0");
    } else {
        o.put(digits, (size_t)decpt);
        o.put('.');
        o.put(digits + decpt, (size_t)(nd - decpt));
    }
}

// repr(round(x, 3)) for |x| < 1e12: the correctly rounded 3-decimal string without its
// trailing zeros IS the shortest repr of the double round() returns
void put_round3(Out& o, double x)
{
    char buf[48];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed, 3);
    char* e = r.ptr;
    while (e[-1] == '0' && e[-2] != '.') e--;
    o.put(buf, (size_t)(e - buf));
}

}  // namespace

extern "C" int64_t pxg_summary_rows(const pxg_summary_columns* c, char* out, int64_t cap)
{
    if (!c || (!out && cap) || c->n < 0) return PXG_E_INVALID;
    Out o{ out, out + cap };
    for (int64_t k = 0; k < c->n; k++) {
        const int64_t b = c->string_row[k];
        for (int f = 0; f < 4; f++) {            // filename, read_id, run_id, channel
            if (!put_ucs4(o, c->text[f].data + b * c->text[f].width, c->text[f].width)) return PXG_E_UNSUPPORTED;
            o.put('\t');
        }
        const double start = (double)c->start_time[k] / c->sampling_rate[k];
        if (!(std::fabs(start) < 1e12)) return PXG_E_UNSUPPORTED;
        put_round3(o, start);
        o.put('\t');
        o.put_int(c->duration[k]);
        o.put('\t');
        o.put_int(c->num_events[k]);
        o.put('\t');
        o.put_int(c->sequence_length[k]);
        o.put('\t');
        if (c->has_summary[k]) put_repr(o, c->mean_qscore[k]);
        else o.put('0');                         // report(): the int 0 until a basecall summary is loaded
        o.put('\t');
        if (!put_ucs4(o, c->text[4].data + b * c->text[4].width, c->text[4].width)) return PXG_E_UNSUPPORTED;   // sample_id
        o.put('\t');
        if (c->status[k] < 0 || c->status[k] >= c->n_status || c->label[k] < 0 || c->label[k] >= c->n_labels)
            return PXG_E_INVALID;
        o.put(c->status_names[c->status[k]]);
        o.put('\t');
        o.put(c->label_names[c->label[k]]);
        if (c->barcode_names) {                  // barcoding on: name ([0] = no call) and score
            const int bc = c->barcode[k] + 1;
            if (bc < 0 || bc >= c->n_barcode_names) return PXG_E_INVALID;
            o.put('\t');
            o.put(c->barcode_names[bc]);
            o.put('\t');
            o.put_int(bc ? c->barcode_score[k] : 0);
        }
        if (c->polya_dwell) {                    // measure_polya on: '%.4f' or empty
            o.put('\t');
            if (c->has_polya[k]) {
                char buf[48];
                auto r = std::to_chars(buf, buf + sizeof buf, c->polya_dwell[k], std::chars_format::fixed, 4);
                o.put(buf, (size_t)(r.ptr - buf));
            }
        }
        o.put('\n');
        if (!o.ok) return PXG_E_NOMEM;           // the caller's buffer is too small
    }
    return (int64_t)(o.p - out);
}
