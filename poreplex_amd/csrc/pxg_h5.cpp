// pxg_h5.cpp -- libpxghost.so, host only: FAST5 (HDF5) input without an HDF5 library.
//
// The reference opens every read with h5py (poreplex/fast5_file.py:37-58 get_read_ids, :97-131
// Fast5Reader: attributes of Raw / channel_id / tracking_id, the int16 `Signal' dataset, :133-181
// the basecall group).  One h5py round trip per attribute and per read is a few hundred
// microseconds of Python -- nothing next to the reference's own per-read cost, everything next to a GPU
// path that needs 1.5 us per read.  This file reads the subset of the HDF5 file format FAST5 files
// are written in, straight from a memory map, and decodes the signals of a whole batch on host
// threads into the page-locked staging arena the GPU copies from:
//   * superblock 0 / 1 / 2 / 3; object headers v1 and v2 (continuation blocks); groups as symbol
//     tables (v1 B-tree + SNOD + local heap: h5py's / the HDF5 library's default, `libver earliest')
//     and as compact link messages; attribute messages v1-3 (fixed-point, float, fixed and
//     variable-length strings through the global heap); datasets compact / contiguous / chunked
//     (v1 chunk B-tree; layout v4 single-chunk and implicit index); filters: deflate (zlib),
//     shuffle, fletcher32, and ONT's VBZ (id 32020: zstd + 16-bit streamvbyte of zig-zag deltas;
//     libzstd is taken with dlopen -- absent library = that file's reads fail loudly);
//   * dense link / attribute storage (fractal heaps: files written with `libver latest' and more
//     than eight links in a group) is NOT read: PXG_E_UNSUPPORTED with a message, never a guess.
// Every offset that comes out of the file is bounds-checked against the map.
// Format facts restated from the HDF5 File Format Specification 3.0 and ONT's vbz_compression
// README (public documents); nothing here derives from the reference's sources.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#include <tmmintrin.h>
#define PXG_H5_X86 1
#else
#define PXG_H5_X86 0      // (no vector paths: memcpy and the scalar VBZ loop)
#endif

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pxg.h"

namespace {

struct H5Error {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const std::string& m) { throw H5Error{ code, m }; }

const uint64_t UNDEF = ~0ull;

struct Datatype {
    int cls = -1;              // 0 int, 1 float, 3 string, 6 compound, 9 vlen
    uint32_t size = 0;
    bool is_signed = false;
    bool vlen_string = false;
    struct Member { std::string name; uint32_t offset; int cls; uint32_t size; bool is_signed; };
    std::vector<Member> members;   // compound
};

struct Filter { uint16_t id; std::vector<uint32_t> cd; };

struct Dataset {
    Datatype type;
    std::vector<uint64_t> dims;    // empty: scalar
    int layout = -1;               // 0 compact, 1 contiguous, 2 chunked (v1 B-tree), 3 single chunk, 4 implicit chunks
    uint64_t addr = UNDEF, size = 0;
    const uint8_t* compact = nullptr;
    std::vector<uint64_t> chunk;   // chunk dims (elements), without the element-size dimension
    uint64_t single_size = 0;      // filtered size of a single chunk
    std::vector<Filter> filters;
    uint64_t n_elements() const
    {
        uint64_t n = 1;
        for (uint64_t d : dims) {
            if (d && n > (~0ull) / d) return ~0ull;          // overflow: no size check downstream lets this pass
            n *= d;
        }
        return n;
    }
};

struct Attr {
    Datatype type;
    uint64_t n = 1;
    const uint8_t* data = nullptr;
    size_t len = 0;
};

struct Object {
    std::map<std::string, Attr> attrs;
    bool is_group = false;
    uint64_t btree = UNDEF, heap = UNDEF;                 // symbol-table group
    std::vector<std::pair<std::string, uint64_t>> links;  // compact links
    bool has_dataset = false;
    Dataset ds;
};

typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
struct Zstd {
    zstd_decompress_fn decompress = nullptr;
    zstd_iserror_fn is_error = nullptr;
    // (optional: a decompression context kept per thread saves the workspace set-up of every call)
    void* (*create_dctx)() = nullptr;
    size_t (*free_dctx)(void*) = nullptr;
    size_t (*decompress_dctx)(void*, void*, size_t, const void*, size_t) = nullptr;
    Zstd()
    {
        for (const char* name : { "libzstd.so.1", "libzstd.so" }) {
            void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            decompress = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
            is_error = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
            create_dctx = (void* (*)())dlsym(h, "ZSTD_createDCtx");
            free_dctx = (size_t (*)(void*))dlsym(h, "ZSTD_freeDCtx");
            decompress_dctx = (size_t (*)(void*, void*, size_t, const void*, size_t))dlsym(h, "ZSTD_decompressDCtx");
            if (!create_dctx || !free_dctx || !decompress_dctx) create_dctx = nullptr;
            if (decompress && is_error) return;
        }
        decompress = nullptr;
    }
    size_t run(void* dst, size_t cap, const void* src, size_t n) const;
};
const Zstd& zstd()
{
    static Zstd z;
    return z;
}
struct ZstdThread {
    void* ctx = nullptr;
    ~ZstdThread() { if (ctx) zstd().free_dctx(ctx); }
};
size_t Zstd::run(void* dst, size_t cap, const void* src, size_t n) const
{
    if (create_dctx) {
        static thread_local ZstdThread t;
        if (!t.ctx) t.ctx = create_dctx();
        if (t.ctx) return decompress_dctx(t.ctx, dst, cap, src, n);
    }
    return decompress(dst, cap, src, n);
}

// ---- VBZ version 1: eight 16-bit samples per control byte, sample q is 1 + bit q bytes long ----------------
// One pshufb spreads the 8..16 data bytes of a group over eight 16-bit lanes (a 256 x 16-byte table of byte
// positions), zig-zag and the running sum stay in the vector: ~12 instructions per 8 samples where the scalar
// loop takes ~50.  Same integers as the scalar loop (sums modulo 2^16); PXG_H5_SCALAR=1 or a CPU without SSSE3
// keeps the scalar loop.
struct SvbTable {
    alignas(16) uint8_t shuf[256][16];
    uint8_t len[256];
    SvbTable()
    {
        for (int k = 0; k < 256; k++) {
            int at = 0;
            for (int q = 0; q < 8; q++) {
                const int two = (k >> q) & 1;
                shuf[k][2 * q] = (uint8_t)at;
                shuf[k][2 * q + 1] = two ? (uint8_t)(at + 1) : 0x80;
                at += 1 + two;
            }
            len[k] = (uint8_t)at;
        }
    }
};
static bool svb_simd_ok()
{
#if PXG_H5_X86
    static const bool ok = !getenv("PXG_H5_SCALAR") && __builtin_cpu_supports("ssse3");
    return ok;
#else
    return false;
#endif
}
// groups [0, n_groups) of `keys` / `data` -> o16; returns the data bytes used, or (size_t)-1 when the stream ends
// inside a group.  `data` must be readable 16 bytes past every group start (the caller pads its buffer).
#if PXG_H5_X86
__attribute__((target("ssse3")))
static size_t svb1_groups_ssse3(const uint8_t* keys, const uint8_t* data, size_t n_data, uint64_t n_groups, bool zig,
                                uint16_t* o16, uint32_t& prev_io)
{
    static const SvbTable T;
    size_t pos = 0;
    __m128i prev = _mm_set1_epi16((short)prev_io);
    const __m128i one = _mm_set1_epi16(1), zero = _mm_setzero_si128();
    const __m128i last = _mm_set_epi8(15, 14, 15, 14, 15, 14, 15, 14, 15, 14, 15, 14, 15, 14, 15, 14);
    for (uint64_t g = 0; g < n_groups; g++) {
        const unsigned key = keys[g];
        const size_t group = T.len[key];
        if (pos + group > n_data) return (size_t)-1;
        __m128i v = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + pos)), _mm_load_si128((const __m128i*)T.shuf[key]));
        if (zig) v = _mm_xor_si128(_mm_srli_epi16(v, 1), _mm_sub_epi16(zero, _mm_and_si128(v, one)));
        v = _mm_add_epi16(v, _mm_slli_si128(v, 2));
        v = _mm_add_epi16(v, _mm_slli_si128(v, 4));
        v = _mm_add_epi16(v, _mm_slli_si128(v, 8));
        v = _mm_add_epi16(v, prev);
        _mm_storeu_si128((__m128i*)(o16 + 8 * g), v);
        prev = _mm_shuffle_epi8(v, last);
        pos += group;
    }
    prev_io = (uint32_t)(uint16_t)_mm_extract_epi16(prev, 0);
    return pos;
}
#else
static size_t svb1_groups_ssse3(const uint8_t*, const uint8_t*, size_t, uint64_t, bool, uint16_t*, uint32_t&) { return (size_t)-1; }
#endif

}  // namespace

struct pxg_h5_read {               // one read of an open file
    std::string id;                // read_<id> group name without the prefix / the attribute
    uint64_t raw_obj = UNDEF;      // object header of the group that carries the Raw attributes
    uint64_t signal_obj = UNDEF, channel_obj = UNDEF, tracking_obj = UNDEF, analyses_obj = UNDEF;
};

// Where a read's basecall text and move table lie in the file, noted by the metadata pass (pxg_h5_info has
// walked Analyses/Basecall_1D_xxx, read and validated the Fastq record and summed the moves): the batch decoder
// then copies three byte ranges instead of walking five groups again.  Only for the plain layouts (fixed-length
// Fastq string and uint8 Move table stored contiguously, unfiltered); everything else takes the walk.
struct BcWhere {
    std::atomic<int> state{ 0 };   // 0 unknown, 2 being written, 1 usable
    uint64_t seq_at = 0, qual_at = 0, len = 0, move_at = UNDEF, n_moves = 0;
};

struct pxg_h5 {
    mutable std::unique_ptr<BcWhere[]> bc_where;      // one per read (allocated when the reads are known)
    int fd = -1;
    const uint8_t* p = nullptr;
    size_t n = 0, map_len = 0;
    static constexpr size_t SLACK = 1 << 16;     // zero bytes mapped behind the file
    uint64_t base = 0;
    int so = 8, sl = 8;            // size of offsets / lengths
    uint64_t root = UNDEF;
    bool multi = false;
    std::vector<pxg_h5_read> reads;
    std::string err;

    const uint8_t* at(uint64_t off, uint64_t len) const
    {
        if (off == UNDEF || off + base < off || off + base > n || len > n - (off + base))
            fail(PXG_E_INVALID, "HDF5: a structure points outside the file (truncated or corrupt)");
        return p + base + off;
    }
    // `len` raw bytes of the file at `off` into `out`.  A big stretch (a read's uncompressed Signal: ~120 KB) is
    // fetched with pread() instead of a memcpy from the mapping, which takes a minor fault and a TLB fill for every
    // 4 KB page it has not touched before -- 300 000 of them per 10 000-read batch, on all loader threads at once,
    // against the same address space the group walk of the NEXT file is faulting in.  The copy is memory-bound
    // (16-core quota of the GPU box: 46-52 GB/s), so the default goes through a 256 KB buffer that stays in the
    // core's L2 and leaves with non-temporal stores: the destination is never read for ownership, two passes over
    // memory instead of three.  Loader call for 10 000 reads (profiles/r05/fast5_ingest_*.txt, ab_h5_copy.txt):
    // memcpy 62 ms, nt (non-temporal stores from the mapping: the faults stay) 64, pread 52; with the reader's
    // shared worker threads pread 31.9, bounce 27.7.
    // PXG_H5_COPY=memcpy|pread|nt|bounce selects the path (A/B; default bounce).
    static int copy_mode()
    {
        static const int mode = [] {
            const char* e = getenv("PXG_H5_COPY");
            return !e ? 3 : (!strcmp(e, "memcpy") ? 0 : (!strcmp(e, "nt") ? 2 : (!strcmp(e, "pread") ? 1 : 3)));
        }();
        return mode;
    }
    void copy_out(uint8_t* out, uint64_t off, uint64_t len) const
    {
        const uint8_t* src = at(off, len);
        int mode = len >= (64u << 10) ? copy_mode() : 0;
#if !PXG_H5_X86
        if (mode >= 2) mode = 1;                       // the non-temporal paths are x86 code: plain pread instead
#endif
        if (mode == 1 && fd >= 0) {
            uint64_t done = 0;
            while (done < len) {
                const ssize_t got = pread(fd, out + done, (size_t)(len - done), (off_t)(base + off + done));
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) break;                       // (a file shrinking under the map: take the mapping's zeros)
                done += (uint64_t)got;
            }
            if (done < len) memcpy(out + done, src + done, (size_t)(len - done));
            return;
        }
#if PXG_H5_X86
        if (mode == 3 && fd >= 0) {
            // pread into a buffer that stays in this core's L2, then non-temporal stores: the destination is
            // never read for ownership (two passes over memory instead of three)
            static thread_local std::vector<uint8_t> bounce;
            const size_t CH = 256u << 10;
            if (bounce.size() < CH + 64) bounce.resize(CH + 64);
            uint8_t* b = (uint8_t*)(((uintptr_t)bounce.data() + 63) & ~(uintptr_t)63);
            uint64_t done = 0;
            bool ok = true;
            while (done < len && ok) {
                // keep (out + done) 16-byte aligned after the first piece
                size_t want = (size_t)std::min<uint64_t>(CH, len - done);
                if (done == 0) want = std::min<size_t>(want, ((16 - ((uintptr_t)out & 15)) & 15) + (CH - 16));
                size_t have = 0;
                while (have < want) {
                    const ssize_t got = pread(fd, b + have, want - have, (off_t)(base + off + done + have));
                    if (got < 0 && errno == EINTR) continue;
                    if (got <= 0) { ok = false; break; }
                    have += (size_t)got;
                }
                if (!ok) break;
                uint8_t* o = out + done;
                size_t i = std::min<size_t>(have, (size_t)((16 - ((uintptr_t)o & 15)) & 15));
                memcpy(o, b, i);
                for (; i + 64 <= have; i += 64) {
                    const __m128i x0 = _mm_loadu_si128((const __m128i*)(b + i)), x1 = _mm_loadu_si128((const __m128i*)(b + i + 16));
                    const __m128i x2 = _mm_loadu_si128((const __m128i*)(b + i + 32)), x3 = _mm_loadu_si128((const __m128i*)(b + i + 48));
                    _mm_stream_si128((__m128i*)(o + i), x0);
                    _mm_stream_si128((__m128i*)(o + i + 16), x1);
                    _mm_stream_si128((__m128i*)(o + i + 32), x2);
                    _mm_stream_si128((__m128i*)(o + i + 48), x3);
                }
                memcpy(o + i, b + i, have - i);
                done += have;
            }
            _mm_sfence();
            if (done < len) memcpy(out + done, src + done, (size_t)(len - done));
            return;
        }
        if (mode == 2) {
            const size_t head = (size_t)((16 - ((uintptr_t)out & 15)) & 15);
            memcpy(out, src, head);
            size_t i = head;
            for (; i + 64 <= len; i += 64) {
                const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
                const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
                _mm_stream_si128((__m128i*)(out + i), a);
                _mm_stream_si128((__m128i*)(out + i + 16), b);
                _mm_stream_si128((__m128i*)(out + i + 32), c);
                _mm_stream_si128((__m128i*)(out + i + 48), d);
            }
            _mm_sfence();
            memcpy(out + i, src + i, (size_t)len - i);
            return;
        }
#endif
        memcpy(out, src, (size_t)len);
    }
    static uint64_t rd(const uint8_t* q, int bytes)
    {
        uint64_t v = 0;
        for (int i = bytes - 1; i >= 0; i--) v = (v << 8) | q[i];
        return v;
    }
    uint64_t off_at(const uint8_t* q) const         // an address; all ones = undefined
    {
        const uint64_t v = rd(q, so);
        return (so < 8 && v == ((1ull << (8 * so)) - 1)) ? UNDEF : v;
    }
    uint64_t len_at(const uint8_t* q) const { return rd(q, sl); }

    // ---- datatypes / dataspaces ------------------------------------------------------
    size_t parse_datatype(const uint8_t* q, size_t avail, Datatype& t, int depth = 0) const
    {
        if (avail < 8 || depth > 4) fail(PXG_E_INVALID, "HDF5: datatype message too short");
        const int cls = q[0] & 15, ver = q[0] >> 4;
        const uint32_t bits = q[1] | (q[2] << 8) | (q[3] << 16);
        t.cls = cls;
        t.size = (uint32_t)rd(q + 4, 4);
        size_t used = 8;
        switch (cls) {
        case 0:
            if (bits & 1) fail(PXG_E_UNSUPPORTED, "HDF5: big-endian integers are not read");
            t.is_signed = (bits >> 3) & 1; used += 4; break;
        case 1:
            if (bits & 1) fail(PXG_E_UNSUPPORTED, "HDF5: big-endian floats are not read");
            used += 12; break;
        case 3: break;
        case 9: {
            t.vlen_string = (bits & 15) == 1;
            Datatype basetype;
            used += parse_datatype(q + used, avail - used, basetype, depth + 1);
            break;
        }
        case 6: {
            const int nm = bits & 0xFFFF;
            for (int m = 0; m < nm; m++) {
                Datatype::Member mem;
                const char* name = (const char*)q + used;
                const size_t nl = strnlen(name, avail - used);
                if (nl >= avail - used) fail(PXG_E_INVALID, "HDF5: compound member name runs off the message");
                mem.name.assign(name, nl);
                used += ver < 3 ? ((nl + 1 + 7) & ~(size_t)7) : nl + 1;
                if (ver < 3) {
                    mem.offset = (uint32_t)rd(q + used, 4);
                    used += 4;
                    if (ver == 1) used += 1 + 3 + 4 + 4 + 16;
                } else {
                    int ob = 1;
                    while (ob < 4 && (t.size >> (8 * ob))) ob++;
                    mem.offset = (uint32_t)rd(q + used, ob);
                    used += ob;
                }
                if (used > avail) fail(PXG_E_INVALID, "HDF5: compound datatype runs off the message");
                Datatype mt;
                used += parse_datatype(q + used, avail - used, mt, depth + 1);
                mem.cls = mt.cls; mem.size = mt.size; mem.is_signed = mt.is_signed;
                t.members.push_back(mem);
            }
            break;
        }
        default:
            // other classes (time, bitfield, opaque, reference, enum, array): sizes only; the
            // FAST5 fields this reader serves never use them
            if (cls == 8 || cls == 10) fail(PXG_E_UNSUPPORTED, "HDF5: enum / array datatypes are not read");
            break;
        }
        if (used > avail) fail(PXG_E_INVALID, "HDF5: datatype runs off the message");
        return used;
    }

    void parse_dataspace(const uint8_t* q, size_t avail, std::vector<uint64_t>& dims) const
    {
        if (avail < 4) fail(PXG_E_INVALID, "HDF5: dataspace message too short");
        const int ver = q[0], rank = q[1];
        size_t at_ = ver == 1 ? 8 : 4;
        if (ver == 2 && q[3] == 2) { dims.assign(1, 0); return; }      // null dataspace
        dims.clear();
        if (at_ + (size_t)rank * sl > avail) fail(PXG_E_INVALID, "HDF5: dataspace runs off the message");
        for (int d = 0; d < rank; d++) dims.push_back(len_at(q + at_ + (size_t)d * sl));
    }

    // ---- object headers -----------------------------------------------------------------
    void message(Object& o, int type, const uint8_t* q, size_t size, std::vector<std::pair<uint64_t, uint64_t>>& more) const
    {
        switch (type) {
        case 0x0001: parse_dataspace(q, size, o.ds.dims); o.has_dataset = true; break;
        case 0x0003: parse_datatype(q, size, o.ds.type); break;
        case 0x0008: {
            Dataset& d = o.ds;
            const int ver = q[0];
            if (ver == 3) {
                const int cls = q[1];
                if (cls == 0) { d.layout = 0; d.size = rd(q + 2, 2); d.compact = q + 4; if (4 + d.size > size) fail(PXG_E_INVALID, "HDF5: compact data runs off the message"); }
                else if (cls == 1) { d.layout = 1; d.addr = off_at(q + 2); d.size = len_at(q + 2 + so); }
                else if (cls == 2) {
                    d.layout = 2;
                    const int nd = q[2];
                    d.addr = off_at(q + 3);
                    d.chunk.clear();
                    if (3 + (size_t)so + 4 * (size_t)nd > size) fail(PXG_E_INVALID, "HDF5: layout message too short");
                    for (int k = 0; k + 1 < nd; k++) d.chunk.push_back(rd(q + 3 + so + 4 * k, 4));
                } else fail(PXG_E_UNSUPPORTED, "HDF5: virtual datasets are not read");
            } else if (ver == 4) {
                const int cls = q[1];
                if (cls == 0) { d.layout = 0; d.size = rd(q + 2, 2); d.compact = q + 4; if (4 + d.size > size) fail(PXG_E_INVALID, "HDF5: compact data runs off the message"); }
                else if (cls == 1) { d.layout = 1; d.addr = off_at(q + 2); d.size = len_at(q + 2 + so); }
                else if (cls == 2) {
                    const int flags = q[2], nd = q[3], enc = q[4];
                    size_t a = 5;
                    if (enc < 1 || enc > 8 || a + (size_t)nd * enc + 1 + 2 * (size_t)sl + so > size + 16) fail(PXG_E_INVALID, "HDF5: layout message too short");
                    d.chunk.clear();
                    for (int k = 0; k < nd; k++, a += enc)
                        if (k + 1 < nd) d.chunk.push_back(rd(q + a, enc));
                    const int index = q[a++];
                    if (index == 1) {
                        d.layout = 3;
                        if (flags & 2) { d.single_size = len_at(q + a); a += sl + 4; }
                        d.addr = off_at(q + a);
                    } else if (index == 2) {
                        d.layout = 4;
                        d.addr = off_at(q + a);
                    } else
                        fail(PXG_E_UNSUPPORTED, "HDF5: chunk index (fixed / extensible array, v2 B-tree) of a file "
                                                "written with `libver latest' is not read");
                } else fail(PXG_E_UNSUPPORTED, "HDF5: virtual datasets are not read");
            } else if (ver == 1 || ver == 2) {
                const int nd = q[1], cls = q[2];
                size_t a = 8;
                if (cls != 0) { d.addr = off_at(q + a); a += so; }
                std::vector<uint64_t> dd;
                if (a + 4 * (size_t)nd + 4 > size + 8) fail(PXG_E_INVALID, "HDF5: layout message too short");
                for (int k = 0; k < nd; k++, a += 4) dd.push_back(rd(q + a, 4));
                if (cls == 1) { d.layout = 1; d.size = 0; }
                else if (cls == 2) { d.layout = 2; d.chunk.assign(dd.begin(), dd.end() - (dd.empty() ? 0 : 1)); }
                else { d.layout = 0; d.size = rd(q + a, 4); d.compact = q + a + 4; if (a + 4 + d.size > size) fail(PXG_E_INVALID, "HDF5: compact data runs off the message"); }
            } else fail(PXG_E_UNSUPPORTED, "HDF5: unknown data layout message version");
            break;
        }
        case 0x000B: {
            const int ver = q[0], nf = q[1];
            size_t a = ver == 1 ? 8 : 2;
            o.ds.filters.clear();
            for (int f = 0; f < nf; f++) {
                if (a + 8 > size + 8) fail(PXG_E_INVALID, "HDF5: filter pipeline runs off the message");
                Filter fl;
                fl.id = (uint16_t)rd(q + a, 2); a += 2;
                size_t name_len = 0;
                if (ver == 1 || fl.id >= 256) { name_len = rd(q + a, 2); a += 2; }
                a += 2;                                               // flags
                const int ncd = (int)rd(q + a, 2); a += 2;
                a += ver == 1 ? ((name_len + 7) & ~(size_t)7) : name_len;
                if (a + 4 * (size_t)ncd > size) fail(PXG_E_INVALID, "HDF5: filter pipeline runs off the message");
                for (int c = 0; c < ncd; c++, a += 4) fl.cd.push_back((uint32_t)rd(q + a, 4));
                if (ver == 1 && (ncd & 1)) a += 4;
                if (a > size) fail(PXG_E_INVALID, "HDF5: filter pipeline runs off the message");
                o.ds.filters.push_back(fl);
            }
            break;
        }
        case 0x000C: {
            const int ver = q[0];
            if (ver < 1 || ver > 3) fail(PXG_E_UNSUPPORTED, "HDF5: unknown attribute message version");
            if (ver >= 2 && (q[1] & 3)) fail(PXG_E_UNSUPPORTED, "HDF5: attributes with shared datatypes are not read");
            const size_t nl = rd(q + 2, 2), tl = rd(q + 4, 2), sl_ = rd(q + 6, 2);
            size_t a = ver == 3 ? 9 : 8;
            auto pad = [&](size_t v) { return ver == 1 ? ((v + 7) & ~(size_t)7) : v; };
            if (a + pad(nl) + pad(tl) + pad(sl_) > size) fail(PXG_E_INVALID, "HDF5: attribute runs off its message");
            std::string name((const char*)q + a, strnlen((const char*)q + a, nl));
            a += pad(nl);
            Attr at_;
            parse_datatype(q + a, tl, at_.type);
            a += pad(tl);
            std::vector<uint64_t> dims;
            parse_dataspace(q + a, sl_, dims);
            a += pad(sl_);
            at_.data = q + a;
            at_.len = size - a;
            at_.n = 1;
            for (uint64_t d : dims) {        // (the product is checked while it is formed: it cannot wrap past the test below)
                if (d && at_.n > at_.len / d + 1) fail(PXG_E_INVALID, "HDF5: attribute dimensions exceed its message");
                at_.n *= d;
            }
            if ((uint64_t)at_.type.size * at_.n > at_.len) fail(PXG_E_INVALID, "HDF5: attribute data runs off its message");
            o.attrs[name] = at_;
            break;
        }
        case 0x0010: more.push_back({ off_at(q), len_at(q + so) }); break;
        case 0x0011: o.is_group = true; o.btree = off_at(q); o.heap = off_at(q + so); break;
        case 0x0002: {
            o.is_group = true;
            const int flags = q[1];
            size_t a = 2 + ((flags & 1) ? 8 : 0);
            if (off_at(q + a) != UNDEF)
                fail(PXG_E_UNSUPPORTED, "HDF5: group with dense link storage (fractal heap: a file written with `libver "
                                        "latest' and more than eight links in a group) is not read");
            break;
        }
        case 0x0006: {
            const int flags = q[1];
            size_t a = 2;
            int ltype = 0;
            if (flags & 8) ltype = q[a++];
            if (flags & 4) a += 8;
            if (flags & 16) a += 1;
            const int lb = 1 << (flags & 3);
            const size_t nl = rd(q + a, lb); a += lb;
            if (a + nl > size) fail(PXG_E_INVALID, "HDF5: link name runs off its message");
            std::string name((const char*)q + a, nl);
            a += nl;
            if (ltype == 0) o.links.push_back({ name, off_at(q + a) });
            o.is_group = true;
            break;
        }
        case 0x0015: {
            const int flags = q[1];
            size_t a = 2 + ((flags & 1) ? 2 : 0);
            if (off_at(q + a) != UNDEF)
                fail(PXG_E_UNSUPPORTED, "HDF5: object with dense attribute storage (fractal heap) is not read");
            break;
        }
        default: break;
        }
    }

    Object object(uint64_t addr) const
    {
        Object o;
        const uint8_t* h = at(addr, 16);
        std::vector<std::pair<uint64_t, uint64_t>> more;
        if (memcmp(h, "OHDR", 4) == 0) {                       // version 2
            const int flags = h[5];
            size_t a = 6 + ((flags & 32) ? 16 : 0) + ((flags & 16) ? 4 : 0);
            const int cb = 1 << (flags & 3);
            h = at(addr, a + cb);
            const uint64_t csize = rd(h + a, cb);
            a += cb;
            auto walk = [&](const uint8_t* q, uint64_t len) {
                uint64_t pos = 0;
                const size_t hdr = 4 + ((flags & 4) ? 2 : 0);
                while (pos + hdr <= len) {
                    const int type = q[pos];
                    const size_t sz = rd(q + pos + 1, 2);
                    if (pos + hdr + sz > len) break;           // gap before the checksum
                    message(o, type, q + pos + hdr, sz, more);
                    pos += hdr + sz;
                }
            };
            walk(at(addr + a, csize + 4), csize);
            for (size_t k = 0; k < more.size() && k < 4096; k++) {
                const uint8_t* c = at(more[k].first, more[k].second);
                if (more[k].second < 8 || memcmp(c, "OCHK", 4)) fail(PXG_E_INVALID, "HDF5: bad object header continuation");
                walk(c + 4, more[k].second - 8);
            }
            return o;
        }
        if (h[0] != 1) fail(PXG_E_INVALID, "HDF5: unknown object header version");
        int n_msgs = (int)rd(h + 2, 2);
        const uint64_t hsize = rd(h + 8, 4);
        auto walk = [&](const uint8_t* q, uint64_t len) {
            uint64_t pos = 0;
            while (n_msgs > 0 && pos + 8 <= len) {
                const int type = (int)rd(q + pos, 2);
                const size_t sz = rd(q + pos + 2, 2);
                const int mflags = q[pos + 4];
                if (pos + 8 + sz > len) fail(PXG_E_INVALID, "HDF5: header message runs off its block");
                if ((mflags & 2) && type != 0x0010)
                    fail(PXG_E_UNSUPPORTED, "HDF5: shared header messages (committed datatypes) are not read");
                message(o, type, q + pos + 8, sz, more);
                pos += 8 + sz;
                n_msgs--;
            }
        };
        walk(at(addr + 16, hsize), hsize);
        for (size_t k = 0; k < more.size() && k < 4096 && n_msgs > 0; k++)
            walk(at(more[k].first, more[k].second), more[k].second);
        return o;
    }

    // ---- groups -------------------------------------------------------------------------------
    // every B-tree walk spends from a node budget: a file of n bytes holds at most n / 24 nodes, and a node
    // that names itself (or an ancestor) as its child would otherwise cost fan-out ^ depth visits
    void spend_node(uint64_t& budget) const
    {
        if (budget == 0) fail(PXG_E_INVALID, "HDF5: B-tree visits more nodes than the file can hold (cyclic?)");
        budget--;
    }
    void group_btree(uint64_t node, const uint8_t* heap_data, uint64_t heap_size, int depth,
                     std::vector<std::pair<std::string, uint64_t>>& out, uint64_t* budget_or_null = nullptr) const
    {
        uint64_t local = (uint64_t)n / 24 + 16;
        uint64_t& budget = budget_or_null ? *budget_or_null : local;
        spend_node(budget);
        if (depth > 32) fail(PXG_E_INVALID, "HDF5: group B-tree too deep");
        const uint8_t* t = at(node, 8 + 2 * so);
        if (memcmp(t, "TREE", 4) == 0) {
            if (t[4] != 0) fail(PXG_E_INVALID, "HDF5: not a group B-tree node");
            const int used = (int)rd(t + 6, 2);
            const uint8_t* e = at(node + 8 + 2 * so, (uint64_t)used * (sl + so) + sl);
            for (int k = 0; k < used; k++)
                group_btree(off_at(e + sl + (size_t)k * (sl + so)), heap_data, heap_size, depth + 1, out, &budget);
            return;
        }
        if (memcmp(t, "SNOD", 4)) fail(PXG_E_INVALID, "HDF5: bad group node signature");
        const int nsym = (int)rd(t + 6, 2);
        const size_t esz = 2 * so + 24;
        const uint8_t* e = at(node + 8, (uint64_t)nsym * esz);
        for (int k = 0; k < nsym; k++) {
            const uint64_t name_off = off_at(e + k * esz), obj = off_at(e + k * esz + so);
            if (name_off >= heap_size) fail(PXG_E_INVALID, "HDF5: link name outside the local heap");
            const char* nm = (const char*)heap_data + name_off;
            out.push_back({ std::string(nm, strnlen(nm, heap_size - name_off)), obj });
        }
    }

    std::vector<std::pair<std::string, uint64_t>> children(const Object& g) const
    {
        if (g.btree == UNDEF) return g.links;
        const uint8_t* hp = at(g.heap, 8 + 2 * sl + so);
        if (memcmp(hp, "HEAP", 4)) fail(PXG_E_INVALID, "HDF5: bad local heap signature");
        const uint64_t hsize = len_at(hp + 8), haddr = off_at(hp + 8 + 2 * sl);
        std::vector<std::pair<std::string, uint64_t>> out;
        group_btree(g.btree, at(haddr, hsize), hsize, 0, out);
        return out;
    }

    uint64_t child(const Object& g, const std::string& name) const
    {
        for (const auto& c : children(g))
            if (c.first == name) return c.second;
        return UNDEF;
    }

    uint64_t resolve(uint64_t from, const std::string& path) const
    {
        uint64_t cur = from;
        size_t a = 0;
        while (a < path.size() && cur != UNDEF) {
            size_t b = path.find('/', a);
            if (b == std::string::npos) b = path.size();
            if (b > a) cur = child(object(cur), path.substr(a, b - a));
            a = b + 1;
        }
        return cur;
    }

    // ---- attribute values ---------------------------------------------------------------------
    std::string attr_string(const Attr& a) const
    {
        if (a.type.cls == 3) {
            const char* s = (const char*)a.data;
            return std::string(s, strnlen(s, a.type.size));
        }
        if (a.type.cls == 9 && a.type.vlen_string) {
            if (a.len < (size_t)4 + so + 4) fail(PXG_E_INVALID, "HDF5: variable-length string reference too short");
            const uint64_t len = rd(a.data, 4), coll = off_at(a.data + 4);
            const uint32_t index = (uint32_t)rd(a.data + 4 + so, 4);
            return global_heap(coll, index, len);
        }
        if (a.type.cls == 0) return std::to_string(attr_int(a));
        fail(PXG_E_UNSUPPORTED, "HDF5: attribute is not a string");
    }
    std::string global_heap(uint64_t coll, uint32_t index, uint64_t len) const
    {
        const uint8_t* g = at(coll, 8 + sl);
        if (memcmp(g, "GCOL", 4)) fail(PXG_E_INVALID, "HDF5: bad global heap signature");
        const uint64_t csize = len_at(g + 8);
        g = at(coll, csize);
        uint64_t pos = 8 + sl;
        while (pos + 8 + sl <= csize) {
            const uint32_t idx = (uint32_t)rd(g + pos, 2);
            const uint64_t osz = len_at(g + pos + 8);
            if (idx == 0) break;
            // (osz comes from the file: compared against what is LEFT, so that a value near 2^64 can neither
            //  wrap the bound check nor make the step below add 0 and the loop spin)
            if (osz > csize - pos - 8 - sl) fail(PXG_E_INVALID, "HDF5: global heap object runs off its collection");
            if (idx == index) {
                const char* s = (const char*)g + pos + 8 + sl;
                return std::string(s, strnlen(s, std::min(len, osz)));
            }
            pos += 8 + sl + ((osz + 7) & ~7ull);
        }
        fail(PXG_E_INVALID, "HDF5: global heap object not found");
    }
    int64_t attr_int(const Attr& a) const
    {
        if (a.type.cls == 0 && a.type.size <= 8) {
            uint64_t v = 0;
            for (int i = (int)a.type.size - 1; i >= 0; i--) v = (v << 8) | a.data[i];
            if (a.type.is_signed && a.type.size < 8 && (v >> (8 * a.type.size - 1))) v |= ~0ull << (8 * a.type.size);
            return (int64_t)v;
        }
        if (a.type.cls == 1) return (int64_t)attr_double(a);
        fail(PXG_E_UNSUPPORTED, "HDF5: attribute is not an integer");
    }
    double attr_double(const Attr& a) const
    {
        if (a.type.cls == 1 && a.type.size == 8) { double d; memcpy(&d, a.data, 8); return d; }
        if (a.type.cls == 1 && a.type.size == 4) { float f; memcpy(&f, a.data, 4); return f; }
        if (a.type.cls == 0) return (double)attr_int(a);
        fail(PXG_E_UNSUPPORTED, "HDF5: attribute is not a number");
    }
    const Attr& need(const Object& o, const char* name) const
    {
        auto it = o.attrs.find(name);
        if (it == o.attrs.end()) fail(PXG_E_INVALID, std::string("FAST5: attribute '") + name + "' is missing");
        return it->second;
    }

    // ---- dataset bytes ------------------------------------------------------------------------
    static void unshuffle(std::vector<uint8_t>& buf, size_t esz)
    {
        if (esz < 2) return;
        const size_t n = buf.size() / esz;
        std::vector<uint8_t> out(buf.size());
        for (size_t b = 0; b < esz; b++)
            for (size_t i = 0; i < n; i++) out[i * esz + b] = buf[b * n + i];
        memcpy(out.data() + n * esz, buf.data() + n * esz, buf.size() - n * esz);
        buf.swap(out);
    }

    // undo the filter pipeline of one chunk: `in` -> out (exactly want bytes)
    void unfilter(const Dataset& d, const uint8_t* in, uint64_t in_len, uint32_t mask, uint8_t* out, uint64_t want) const
    {
        std::vector<uint8_t> cur;
        // (the first stage reads the mapped chunk itself: no copy of the compressed bytes)
        struct View { const uint8_t* p; size_t n; };
        bool first = true;
        auto src = [&]() { return first ? View{ in, (size_t)in_len } : View{ cur.data(), cur.size() }; };
        for (int f = (int)d.filters.size() - 1; f >= 0; f--) {
            if (mask & (1u << f)) continue;
            const Filter& fl = d.filters[f];
            if (first && (fl.id == 2 || fl.id == 3)) { cur.assign(in, in + in_len); first = false; }
            if (fl.id == 1) {                                          // deflate
                std::vector<uint8_t> dst(want ? want : 1);
                for (;;) {
                    uLongf got = (uLongf)dst.size();
                    const int rc = uncompress(dst.data(), &got, src().p, (uLong)src().n);
                    if (rc == Z_OK) { dst.resize(got); break; }
                    if (rc != Z_BUF_ERROR || dst.size() > (1ull << 33)) fail(PXG_E_INVALID, "HDF5: deflate stream is corrupt");
                    dst.resize(dst.size() * 2);
                }
                cur.swap(dst);
                first = false;
            } else if (fl.id == 2) {
                unshuffle(cur, fl.cd.empty() ? d.type.size : fl.cd[0]);
            } else if (fl.id == 3) {
                if (cur.size() < 4) fail(PXG_E_INVALID, "HDF5: fletcher32 chunk too short");
                cur.resize(cur.size() - 4);
            } else if (fl.id == 32020) {                               // ONT VBZ
                // cd = (version, integer size, zig-zag, zstd level).  The zstd frame may be preceded by
                // a 4-byte size word (found by the frame's magic number).  Version 1: one control BIT
                // per 16-bit value; version 0: classic streamvbyte over 32-bit widened values (two
                // control bits: 1..4 bytes).  No VBZ file exists in this image: restated from ONT's
                // public description, checked against this build's own writer only (v1).
                const unsigned version = fl.cd.empty() ? 0 : fl.cd[0];
                const unsigned isize = fl.cd.size() > 1 ? fl.cd[1] : 2, zig = fl.cd.size() > 2 ? fl.cd[2] : 1;
                if (isize != 2 || version > 1) fail(PXG_E_UNSUPPORTED, "VBZ: only 16-bit samples, stream versions 0 and 1, are read");
                if (!zstd().decompress) fail(PXG_E_UNSUPPORTED, "VBZ: libzstd.so.1 is not on this host");
                static const uint8_t MAGIC[4] = { 0x28, 0xB5, 0x2F, 0xFD };
                size_t skip = 0;
                const View zin = src();
                if (zin.n >= 8 && memcmp(zin.p, MAGIC, 4) && memcmp(zin.p + 4, MAGIC, 4) == 0) skip = 4;
                const uint64_t n = want / 2;
                // (scratch kept per thread: a 120 KB read would otherwise pay for two zero-filled vectors)
                static thread_local std::vector<uint8_t> svb;
                const size_t svb_cap = (size_t)(version ? (n + 7) / 8 + 2 * n : (n + 3) / 4 + 4 * n);
                if (svb.size() < svb_cap + 32) svb.resize(svb_cap + 32);
                const size_t got = zstd().run(svb.data(), svb_cap, zin.p + skip, zin.n - skip);
                if (zstd().is_error(got)) fail(PXG_E_INVALID, "VBZ: zstd stream is corrupt");
                const size_t keys = version ? (n + 7) / 8 : (n + 3) / 4;
                if (got < keys) fail(PXG_E_INVALID, "VBZ: stream shorter than its control bits");
                memset(svb.data() + got, 0, 32);                    // what a 16-byte load at the last group may see
                // the last stage of the pipeline writes the samples where they belong
                bool last_stage = true;
                for (int g = f - 1; g >= 0; g--) last_stage &= (mask & (1u << g)) != 0;
                std::vector<uint8_t> dst;
                if (!last_stage) dst.resize(want);
                const uint8_t* data = svb.data() + keys;
                const size_t n_data = got - keys;
                uint16_t* o16 = (uint16_t*)(last_stage ? out : dst.data());
                // a stream that turns out corrupt half way must not leave part of a signal in the caller's arena
                // (the read's status says "failed": its slot reads as zeros, as when nothing had been written)
                auto corrupt = [&](const char* what) {
                    if (last_stage) memset(out, 0, (size_t)want);
                    fail(PXG_E_INVALID, what);
                };
                if (last_stage && (want & 1)) out[want - 1] = 0;    // (an odd byte count: the byte no sample covers)
                size_t pos = 0;
                uint32_t prev = 0;
                uint64_t i = 0;
                if (version == 1 && svb_simd_ok()) {
                    pos = svb1_groups_ssse3(svb.data(), data, n_data, n / 8, zig != 0, o16, prev);
                    if (pos == (size_t)-1) corrupt("VBZ: stream ends inside a sample");
                    i = n / 8 * 8;
                } else if (version == 1) {
                    // eight samples per control byte; the byte offsets inside the group come from
                    // the bits below each sample (no data-dependent branch per sample)
                    for (; i + 8 <= n; i += 8) {
                        const unsigned key = svb[i >> 3];
                        const size_t group = 8 + (size_t)__builtin_popcount(key);
                        if (pos + group > n_data) corrupt("VBZ: stream ends inside a sample");
                        const uint8_t* g = data + pos;
                        unsigned at_ = 0;
                        for (unsigned q = 0; q < 8; q++) {
                            const unsigned two = (key >> q) & 1u;
                            uint32_t v = g[at_] | ((uint32_t)(g[at_ + two] & (0u - two)) << 8);
                            at_ += 1 + two;
                            if (zig) v = (v >> 1) ^ (0u - (v & 1));
                            prev += v;
                            o16[i + q] = (uint16_t)prev;
                        }
                        pos += group;
                    }
                }
                for (; i < n; i++) {
                    const unsigned nb = version ? 1 + ((svb[i >> 3] >> (i & 7)) & 1) : 1 + ((svb[i >> 2] >> (2 * (i & 3))) & 3);
                    if (pos + nb > n_data) corrupt("VBZ: stream ends inside a sample");
                    uint32_t v = 0;
                    for (unsigned b = 0; b < nb; b++) v |= (uint32_t)data[pos++] << (8 * b);
                    if (zig) v = (v >> 1) ^ (0u - (v & 1));
                    prev += v;                                   // delta from the previous sample (first: from 0)
                    o16[i] = (uint16_t)prev;
                }
                if (last_stage) return;
                cur.swap(dst);
                first = false;
            } else
                fail(PXG_E_UNSUPPORTED, "HDF5: filter " + std::to_string(fl.id) + " is not read");
        }
        if (src().n < want) fail(PXG_E_INVALID, "HDF5: chunk holds fewer bytes than its dataset");
        memcpy(out, src().p, want);
    }

    void chunk_btree(const Dataset& d, uint64_t node, int depth, uint8_t* out, uint64_t total_el,
                     uint64_t* budget_or_null = nullptr) const
    {
        uint64_t local = (uint64_t)n / 24 + 16;
        uint64_t& budget = budget_or_null ? *budget_or_null : local;
        spend_node(budget);
        if (depth > 32) fail(PXG_E_INVALID, "HDF5: chunk B-tree too deep");
        const int nd = (int)d.chunk.size() + 1;
        const uint8_t* t = at(node, 8 + 2 * so);
        if (memcmp(t, "TREE", 4) || t[4] != 1) fail(PXG_E_INVALID, "HDF5: bad chunk B-tree node");
        const int level = t[5], used = (int)rd(t + 6, 2);
        const size_t ksz = 8 + 8 * (size_t)nd;
        const uint8_t* e = at(node + 8 + 2 * so, (uint64_t)used * (ksz + so) + ksz);
        const uint64_t esz = d.type.size, cel = d.chunk[0];
        for (int k = 0; k < used; k++) {
            const uint8_t* key = e + (size_t)k * (ksz + so);
            const uint64_t child_ = off_at(key + ksz);
            if (level > 0) { chunk_btree(d, child_, depth + 1, out, total_el, &budget); continue; }
            const uint32_t csize = (uint32_t)rd(key, 4), mask = (uint32_t)rd(key + 4, 4);
            const uint64_t first = rd(key + 8, 8);
            if (first >= total_el) continue;
            const uint64_t n_el = std::min(cel, total_el - first);
            if (d.filters.empty()) memcpy(out + first * esz, at(child_, n_el * esz), n_el * esz);
            else if (n_el == cel) unfilter(d, at(child_, csize), csize, mask, out + first * esz, cel * esz);
            else {                                                       // edge chunk: decode whole, keep the head
                std::vector<uint8_t> whole(cel * esz);
                unfilter(d, at(child_, csize), csize, mask, whole.data(), cel * esz);
                memcpy(out + first * esz, whole.data(), n_el * esz);
            }
        }
    }

    // A size that comes out of the file and is about to be allocated: no dataset of this file can
    // hold more than its bytes times the best ratio of the filters read here (deflate: ~1030)
    void sane_bytes(uint64_t n_el, uint64_t esz) const
    {
        const uint64_t cap = (uint64_t)n * 1100 + 65536;
        if (esz == 0 || n_el > cap || esz > cap || n_el * esz > cap)
            fail(PXG_E_INVALID, "HDF5: a dataset claims more bytes than this file could hold (corrupt dimensions)");
    }

    // all bytes of a (1-D or scalar) dataset, fixed-size elements
    void read_dataset(const Dataset& d, uint8_t* out, uint64_t out_bytes) const
    {
        const uint64_t n_el = d.n_elements(), esz = d.type.size, total = n_el * esz;
        sane_bytes(n_el, esz ? esz : 1);
        if (total != out_bytes) fail(PXG_E_INVALID, "HDF5: dataset size differs from what its reader expects");
        if (!total) return;
        if (d.dims.size() > 1) fail(PXG_E_UNSUPPORTED, "HDF5: datasets of more than one dimension are not read");
        switch (d.layout) {
        case 0: if (d.size < total) fail(PXG_E_INVALID, "HDF5: compact dataset too short"); memcpy(out, d.compact, total); return;
        case 1:
            if (d.addr == UNDEF) { memset(out, 0, total); return; }      // never written: fill value 0
            copy_out(out, d.addr, total);
            return;
        case 2:
            memset(out, 0, total);
            if (d.addr == UNDEF) return;
            if (d.chunk.size() != 1 || !d.chunk[0]) fail(PXG_E_UNSUPPORTED, "HDF5: chunked datasets must be one-dimensional");
            sane_bytes(d.chunk[0], esz);
            chunk_btree(d, d.addr, 0, out, n_el);
            return;
        case 3: {
            if (d.chunk.size() != 1) fail(PXG_E_UNSUPPORTED, "HDF5: chunked datasets must be one-dimensional");
            sane_bytes(d.chunk[0], esz);
            const uint64_t cbytes = d.chunk[0] * esz;
            if (d.filters.empty()) { copy_out(out, d.addr, total); return; }
            std::vector<uint8_t> whole(cbytes);
            unfilter(d, at(d.addr, d.single_size), d.single_size, 0, whole.data(), cbytes);
            memcpy(out, whole.data(), std::min(total, cbytes));
            return;
        }
        case 4:
            if (!d.filters.empty()) fail(PXG_E_INVALID, "HDF5: implicit chunk index with filters");
            copy_out(out, d.addr, total);
            return;
        default: fail(PXG_E_INVALID, "HDF5: dataset without a data layout");
        }
    }

    std::string dataset_string(const Dataset& d) const
    {
        if (d.type.cls == 3) {
            sane_bytes(std::max<uint64_t>(d.n_elements(), 1), d.type.size);
            std::vector<uint8_t> buf((size_t)d.type.size * std::max<uint64_t>(d.n_elements(), 1));
            read_dataset(d, buf.data(), buf.size());
            return std::string((const char*)buf.data(), strnlen((const char*)buf.data(), d.type.size));
        }
        if (d.type.cls == 9 && d.type.vlen_string) {
            std::vector<uint8_t> ref(d.type.size);
            Dataset plain = d;
            read_dataset(plain, ref.data(), ref.size());
            return global_heap(off_at(ref.data() + 4), (uint32_t)rd(ref.data() + 4 + so, 4), rd(ref.data(), 4));
        }
        fail(PXG_E_UNSUPPORTED, "HDF5: dataset is not a string");
    }
};

// ------------------------------------------------------------------------------------------------
// C ABI (declared in include/pxg.h)
// ------------------------------------------------------------------------------------------------
static thread_local std::string t_h5_error;

#define H5_GUARD_BEGIN try {
#define H5_GUARD_END(h)                                                              \
    } catch (const H5Error& e) {                                                     \
        t_h5_error = e.msg;                                                          \
        return e.code;                                                               \
    } catch (const std::exception& e) {                                              \
        t_h5_error = std::string("HDF5 reader: ") + e.what();                        \
        return PXG_E_NOMEM;                                                          \
    }

extern "C" const char* pxg_h5_last_error(void) { return t_h5_error.c_str(); }

extern "C" void pxg_h5_close(pxg_h5* h)
{
    if (!h) return;
    if (h->p) munmap((void*)h->p, h->map_len);
    if (h->fd >= 0) close(h->fd);
    delete h;
}

template <typename Fn>
static void run_pool(int64_t n, int threads, Fn fn, int64_t grain = 1);

// `threads`: host threads that walk the read groups of a multi-read file (a 4 000-read file is
// 4 000 groups of four children each: 7 ms on one thread)
extern "C" int pxg_h5_open_mt(const char* path, int32_t threads, pxg_h5** out)
{
    if (!path || !out) return PXG_E_INVALID;
    *out = nullptr;
    pxg_h5* h = new pxg_h5();
    H5_GUARD_BEGIN
    struct Closer { pxg_h5*& h; ~Closer() { if (h) pxg_h5_close(h); } } closer{ h };
    h->fd = open(path, O_RDONLY);
    if (h->fd < 0) fail(PXG_E_INVALID, std::string("Unable to open file '") + path + "' (" + strerror(errno) + ")");
    struct stat st;
    if (fstat(h->fd, &st) || st.st_size < 64) fail(PXG_E_INVALID, std::string("Unable to open file '") + path + "' (file signature not found)");
    h->n = (size_t)st.st_size;
    // The file is mapped over the head of a larger anonymous (zero) region: every structure's START
    // is checked against the file size, and a fixed-size field read near the end of a truncated
    // file then runs into zeros instead of off the map.
    h->map_len = ((h->n + 4095) & ~(size_t)4095) + pxg_h5::SLACK;
    void* m = mmap(nullptr, h->map_len, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { h->p = nullptr; fail(PXG_E_NOMEM, "mmap failed"); }
    h->p = (const uint8_t*)m;
    if (mmap(m, h->n, PROT_READ, MAP_PRIVATE | MAP_FIXED, h->fd, 0) == MAP_FAILED) fail(PXG_E_NOMEM, "mmap failed");
    static const uint8_t SIG[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
    uint64_t sb = UNDEF;
    for (uint64_t o = 0; o + 64 <= h->n; o = o ? o * 2 : 512)
        if (memcmp(h->p + o, SIG, 8) == 0) { sb = o; break; }
    if (sb == UNDEF) fail(PXG_E_INVALID, std::string("Unable to open file '") + path + "' (file signature not found)");
    const uint8_t* s = h->p + sb;
    const int ver = s[8];
    if (ver <= 1) {
        h->so = s[13]; h->sl = s[14];
        size_t a = 24 + (ver == 1 ? 4 : 0);
        if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8)) fail(PXG_E_UNSUPPORTED, "HDF5: odd offset / length sizes");
        h->base = h->off_at(s + a);
        if (h->base == UNDEF) h->base = 0;
        a += 4 * (size_t)h->so;                                  // base, free space, eof, driver info
        h->root = h->off_at(s + a + h->so);                     // root symbol table entry: name offset, header address
    } else if (ver == 2 || ver == 3) {
        h->so = s[9]; h->sl = s[10];
        if ((h->so != 4 && h->so != 8) || (h->sl != 4 && h->sl != 8)) fail(PXG_E_UNSUPPORTED, "HDF5: odd offset / length sizes");
        h->base = h->off_at(s + 12);
        if (h->base == UNDEF) h->base = 0;
        h->root = h->off_at(s + 12 + 3 * (size_t)h->so);
    } else
        fail(PXG_E_UNSUPPORTED, "HDF5: unknown superblock version");
    // ---- the reads of the file (fast5_file.py:37-58, :71-82) ---------------------------------
    const bool trace = getenv("PXG_H5_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    const Object root = h->object(h->root);
    const auto top = h->children(root);
    const auto tr1 = std::chrono::steady_clock::now();
    bool single = false;
    for (const auto& c : top) single |= c.first == "UniqueGlobalKey";
    h->multi = !single;
    if (single) {
        pxg_h5_read r;
        const uint64_t reads = h->resolve(h->root, "Raw/Reads");
        if (reads != UNDEF) {
            auto kids = h->children(h->object(reads));
            std::sort(kids.begin(), kids.end());
            if (!kids.empty()) {
                r.raw_obj = kids[0].second;
                r.signal_obj = h->child(h->object(r.raw_obj), "Signal");
                const Object raw = h->object(r.raw_obj);
                auto it = raw.attrs.find("read_id");
                if (it != raw.attrs.end()) r.id = h->attr_string(it->second);
                r.channel_obj = h->resolve(h->root, "UniqueGlobalKey/channel_id");
                r.tracking_obj = h->resolve(h->root, "UniqueGlobalKey/tracking_id");
                r.analyses_obj = h->child(root, "Analyses");
                if (it != raw.attrs.end()) h->reads.push_back(r);
            }
        }
    } else {
        std::vector<std::pair<std::string, uint64_t>> groups;
        for (const auto& c : top)
            if (c.first.compare(0, 5, "read_") == 0) groups.push_back(c);
        h->reads.resize(groups.size());
        std::atomic<int> failed{ 0 };
        std::string first_error;
        std::mutex err_mu;
        pxg_h5* hh = h;
        run_pool((int64_t)groups.size(), threads, [&](int64_t k) {
            try {
                pxg_h5_read r;
                r.id = groups[(size_t)k].first.substr(5);
                const Object g = hh->object(groups[(size_t)k].second);
                for (const auto& kid : hh->children(g)) {
                    if (kid.first == "Raw") { r.raw_obj = kid.second; r.signal_obj = hh->child(hh->object(kid.second), "Signal"); }
                    else if (kid.first == "channel_id") r.channel_obj = kid.second;
                    else if (kid.first == "tracking_id") r.tracking_obj = kid.second;
                    else if (kid.first == "Analyses") r.analyses_obj = kid.second;
                }
                hh->reads[(size_t)k] = r;
            } catch (const H5Error& e) {
                // a read group that cannot be walked: the read stays listed (its id is known) and
                // fails later, on its own, when its metadata is asked for
                hh->reads[(size_t)k].id = groups[(size_t)k].first.substr(5);
                if (failed.fetch_add(1) == 0) { std::lock_guard<std::mutex> g(err_mu); first_error = e.msg; }
            } catch (const std::exception&) {      // (an allocation: nothing may leave a pool thread)
                hh->reads[(size_t)k] = pxg_h5_read();
                failed.fetch_add(1);
            }
        }, 16);
    }
    if (!getenv("PXG_H5_NO_TEXT_NOTES")) h->bc_where.reset(new BcWhere[h->reads.size() ? h->reads.size() : 1]);
    if (trace)
        fprintf(stderr, "pxg_h5_open: root listing %.2f ms, read groups %.2f ms (%zu reads, %d threads)\n",
                std::chrono::duration<double, std::milli>(tr1 - tr0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr1).count(),
                h->reads.size(), (int)threads);
    *out = h;
    h = nullptr;                     // released to the caller
    return PXG_OK;
    H5_GUARD_END(h)
}

extern "C" int pxg_h5_open(const char* path, pxg_h5** out) { return pxg_h5_open_mt(path, 1, out); }

extern "C" int64_t pxg_h5_n_reads(const pxg_h5* h) { return h ? (int64_t)h->reads.size() : 0; }
extern "C" int pxg_h5_is_multi(const pxg_h5* h) { return h && h->multi; }

static void put(char* dst, size_t cap, const std::string& s, bool may_cut = false)
{
    if (s.size() >= cap && !may_cut)
        fail(PXG_E_UNSUPPORTED, "FAST5: a text attribute is longer than its " + std::to_string(cap - 1) + "-character field");
    memset(dst, 0, cap);
    memcpy(dst, s.data(), std::min(s.size(), cap - 1));
}

static void basecall_of(const pxg_h5* h, const pxg_h5_read& r, pxg_h5_read_info& o, std::string* fastq,
                        std::vector<uint8_t>* moves, std::vector<double>* pms, BcWhere* note = nullptr)
{
    bool plain = false;                      // the text's place can be noted
    uint64_t n_seq_at = 0, n_qual_at = 0, n_len = 0, n_move_at = UNDEF, n_moves_noted = 0;
    o.bc_present = 0; o.bc_table = 0; o.bc_block_stride = 15; o.bc_n_moves = -1; o.bc_move_sum = 0;
    if (r.analyses_obj == UNDEF) return;
    const Object an = h->object(r.analyses_obj);
    std::string best;
    uint64_t best_obj = UNDEF;
    for (const auto& c : h->children(an))
        if (c.first.compare(0, 11, "Basecall_1D") == 0 && (best_obj == UNDEF || c.first > best)) { best = c.first; best_obj = c.second; }
    if (best_obj == UNDEF) return;                                      // fast5_file.py:139-141
    const std::string groupno = best.substr(best.rfind('_') + 1);
    const uint64_t seg = h->resolve(r.analyses_obj, "Segmentation_" + groupno + "/Summary/segmentation");
    if (seg == UNDEF) fail(PXG_E_INVALID, "FAST5: Segmentation_" + groupno + "/Summary/segmentation is missing");
    const Object sego = h->object(seg);
    o.bc_num_events = h->attr_int(h->need(sego, "num_events_template"));
    o.bc_first_sample = h->attr_int(h->need(sego, "first_sample_template"));
    const uint64_t fq = h->resolve(best_obj, "BaseCalled_template/Fastq");
    if (fq == UNDEF) fail(PXG_E_INVALID, "FAST5: BaseCalled_template/Fastq is missing");
    const Dataset fqd = h->object(fq).ds;
    const std::string text = h->dataset_string(fqd);
    // '@name\nSEQ\n+\nQUAL\n'
    size_t l1 = text.find('\n'), l2 = l1 == std::string::npos ? l1 : text.find('\n', l1 + 1);
    size_t l3 = l2 == std::string::npos ? l2 : text.find('\n', l2 + 1);
    if (l3 == std::string::npos) fail(PXG_E_INVALID, "FAST5: Fastq record has fewer than four lines");
    size_t l4 = text.find('\n', l3 + 1);
    if (l4 == std::string::npos) l4 = text.size();
    o.bc_seq_len = (int64_t)(l2 - l1 - 1);
    if (l4 - l3 - 1 != l2 - l1 - 1) fail(PXG_E_INVALID, "FAST5: Fastq sequence and quality lines differ in length");
    for (size_t k = l1 + 1; k < l4; k++)                       // text columns downstream are ASCII by contract
        if (k != l2 && k != l3 && (text[k] < 33 || text[k] > 126) && !(k > l2 && k < l3))
            fail(PXG_E_INVALID, "FAST5: Fastq record holds bytes that are not printable ASCII");
    if (fastq) *fastq = text.substr(l1 + 1, l2 - l1 - 1) + "\n" + text.substr(l3 + 1, l4 - l3 - 1);
    if (fqd.type.cls == 3 && fqd.layout == 1 && fqd.filters.empty() && fqd.addr != UNDEF && fqd.n_elements() <= 1) {
        plain = true;
        n_seq_at = fqd.addr + l1 + 1; n_qual_at = fqd.addr + l3 + 1; n_len = l2 - l1 - 1;
    }
    const uint64_t sm = h->resolve(best_obj, "Summary/basecall_1d_template");
    if (sm == UNDEF) fail(PXG_E_INVALID, "FAST5: Summary/basecall_1d_template is missing");
    const Object smo = h->object(sm);
    o.bc_sequence_length = h->attr_int(h->need(smo, "sequence_length"));
    o.bc_mean_qscore = h->attr_double(h->need(smo, "mean_qscore"));
    auto bs = smo.attrs.find("block_stride");
    if (bs != smo.attrs.end()) o.bc_block_stride = (int32_t)h->attr_int(bs->second);
    o.bc_present = 1;
    // event mapping (fast5_file.py:166-181): `Events' wins over `Move'
    const uint64_t tmpl = h->child(h->object(best_obj), "BaseCalled_template");
    const Object to = h->object(tmpl);
    const uint64_t ev = h->child(to, "Events"), mv = h->child(to, "Move");
    if (ev != UNDEF) {
        plain = false;
        const Dataset d = h->object(ev).ds;
        if (d.type.cls != 6) fail(PXG_E_INVALID, "FAST5: Events is not a table");
        const Datatype::Member* mcol = nullptr;
        const Datatype::Member* pcol = nullptr;
        for (const auto& m : d.type.members) { if (m.name == "move") mcol = &m; if (m.name == "p_model_state") pcol = &m; }
        const size_t ncol = d.type.members.size();
        o.bc_table = (ncol <= 3 && mcol) ? 2 : (ncol == 14 ? 3 : 4);
        if (mcol) {
            const uint64_t n = d.n_elements();
            h->sane_bytes(n, d.type.size);
            if (mcol->size < 1 || mcol->size > 8 || (uint64_t)mcol->offset + mcol->size > d.type.size ||
                (pcol && ((pcol->size != 4 && pcol->size != 8) || (uint64_t)pcol->offset + pcol->size > d.type.size)))
                fail(PXG_E_INVALID, "FAST5: Events table columns lie outside its rows");
            std::vector<uint8_t> rows(n * d.type.size);
            h->read_dataset(d, rows.data(), rows.size());
            o.bc_n_moves = (int64_t)n;
            if (moves) moves->resize(n);
            if (pms && pcol) pms->resize(n);
            for (uint64_t k = 0; k < n; k++) {
                const uint8_t* q = rows.data() + k * d.type.size;
                uint64_t v = 0;
                for (int b = (int)mcol->size - 1; b >= 0; b--) v = (v << 8) | q[mcol->offset + b];
                o.bc_move_sum += (int64_t)(v & 0xFF);
                if (moves) (*moves)[k] = (uint8_t)v;
                if (pms && pcol) {
                    if (pcol->size == 4) { float f; memcpy(&f, q + pcol->offset, 4); (*pms)[k] = f; }
                    else { double f; memcpy(&f, q + pcol->offset, 8); (*pms)[k] = f; }
                }
            }
        }
    } else if (mv != UNDEF) {
        const Dataset d = h->object(mv).ds;
        if (d.type.cls != 0 || d.type.size != 1) fail(PXG_E_INVALID, "FAST5: Move is not a uint8 table");
        const uint64_t n = d.n_elements();
        h->sane_bytes(n, 1);
        std::vector<uint8_t> local;
        std::vector<uint8_t>& buf = moves ? *moves : local;
        buf.resize(n);
        h->read_dataset(d, buf.data(), n);
        o.bc_table = 1;
        o.bc_n_moves = (int64_t)n;
        for (uint64_t k = 0; k < n; k++) o.bc_move_sum += buf[k];
        if (n == 0) n_move_at = UNDEF;
        else if (d.layout == 1 && d.filters.empty() && d.addr != UNDEF) { n_move_at = d.addr; n_moves_noted = n; }
        else plain = false;
    }
    int unknown = 0;
    if (note && plain && note->state.compare_exchange_strong(unknown, 2)) {
        note->seq_at = n_seq_at; note->qual_at = n_qual_at; note->len = n_len;
        note->move_at = n_move_at; note->n_moves = n_moves_noted;
        note->state.store(1, std::memory_order_release);
    }
}

// Metadata text that ends up in ASCII columns downstream (read id, channel number, run id): printable ASCII or the
// read is an error of its own -- a flipped byte must not surface as a decoding exception of a whole batch
// (tools/h5_fuzz.py found that one in round 5).  sample_id is free text: any bytes, decoded leniently by the caller.
static const std::string& ascii_text(const std::string& s, const char* what)
{
    for (unsigned char c : s)
        if (c < 32 || c > 126) fail(PXG_E_INVALID, std::string("FAST5: ") + what + " holds bytes that are not printable ASCII");
    return s;
}

static void info_of(const pxg_h5* h, const pxg_h5_read& r, pxg_h5_read_info& o)
{
    memset(&o, 0, sizeof(o));
    if (r.raw_obj == UNDEF || r.signal_obj == UNDEF || r.channel_obj == UNDEF || r.tracking_obj == UNDEF)
        fail(PXG_E_INVALID, "FAST5: read '" + r.id + "' lacks Raw/Signal, channel_id or tracking_id");
    const Object raw = h->object(r.raw_obj);
    o.duration = h->attr_int(h->need(raw, "duration"));
    o.start_time = h->attr_int(h->need(raw, "start_time"));
    put(o.read_id, sizeof(o.read_id), ascii_text(h->attr_string(h->need(raw, "read_id")), "read_id"));
    const Object ch = h->object(r.channel_obj);
    put(o.channel_number, sizeof(o.channel_number), ascii_text(h->attr_string(h->need(ch, "channel_number")), "channel_number"));
    o.calib.digitisation = h->attr_double(h->need(ch, "digitisation"));
    o.calib.offset = h->attr_double(h->need(ch, "offset"));
    o.calib.range = h->attr_double(h->need(ch, "range"));
    o.calib.sampling_rate = h->attr_double(h->need(ch, "sampling_rate"));
    const Object tr = h->object(r.tracking_obj);
    put(o.run_id, sizeof(o.run_id), ascii_text(h->attr_string(h->need(tr, "run_id")), "run_id"));
    put(o.sample_id, sizeof(o.sample_id), h->attr_string(h->need(tr, "sample_id")));
    const Object sig = h->object(r.signal_obj);
    if (!sig.has_dataset || sig.ds.type.cls != 0 || sig.ds.type.size != 2)
        fail(PXG_E_INVALID, "FAST5: Signal is not a 16-bit integer dataset");
    h->sane_bytes(sig.ds.n_elements(), 2);
    o.n_samples = (int64_t)sig.ds.n_elements();
}

extern "C" int pxg_h5_read_id(const pxg_h5* h, int64_t i, char* out, int64_t cap)
{
    if (!h || i < 0 || i >= (int64_t)h->reads.size() || !out || cap < 2) return PXG_E_INVALID;
    H5_GUARD_BEGIN
    put(out, (size_t)cap, h->reads[(size_t)i].id);
    return PXG_OK;
    H5_GUARD_END(h)
}

// all read ids of the file, '\n'-separated, in one call; returns the bytes needed (call again with a
// buffer of that size when it exceeds cap)
extern "C" int64_t pxg_h5_read_ids(const pxg_h5* h, char* out, int64_t cap)
{
    if (!h) return PXG_E_INVALID;
    int64_t need = 0;
    for (const auto& r : h->reads) need += (int64_t)r.id.size() + 1;
    if (need > cap || !out) return need;
    char* q = out;
    for (const auto& r : h->reads) {
        memcpy(q, r.id.data(), r.id.size());
        q += r.id.size();
        *q++ = '\n';
    }
    return need;
}

// metadata + basecall summary of reads [first, first + n) on `threads` host threads (a 4 000-read
// file is ~60 000 object headers); a read that cannot be described gets status != 0 (its own error,
// as data) and the others are still filled
extern "C" int pxg_h5_info_mt(const pxg_h5* h, int64_t first, int64_t n, pxg_h5_read_info* out, int32_t threads)
{
    if (!h || first < 0 || n < 0 || first + n > (int64_t)h->reads.size() || (n && !out)) return PXG_E_INVALID;
    run_pool(n, threads, [&](int64_t k) {
        pxg_h5_read_info& o = out[k];
        try {
            info_of(h, h->reads[(size_t)(first + k)], o);
            basecall_of(h, h->reads[(size_t)(first + k)], o, nullptr, nullptr, nullptr,
                        h->bc_where ? &h->bc_where[(size_t)(first + k)] : nullptr);
        } catch (const H5Error& e) {
            o.status = e.code;
            put(o.error, sizeof(o.error), e.msg, true);
        } catch (const std::exception& e) {
            o.status = PXG_E_NOMEM;
            put(o.error, sizeof(o.error), e.what(), true);
        }
    }, 8);
    return PXG_OK;
}

extern "C" int pxg_h5_info(const pxg_h5* h, int64_t first, int64_t n, pxg_h5_read_info* out)
{
    return pxg_h5_info_mt(h, first, n, out, 1);
}

// A directory of SINGLE-read files (the reference's classic input: one FAST5 per read) is thousands of opens per
// worker batch; from Python each is a stat, an open, three calls for the read id and one for the metadata -- ~100 us
// of interpreter time per read beside ~40 us of work.  One call opens `n` files on `threads` host threads and, for
// every file that holds exactly one read in the single-read layout, fills that read's pxg_h5_read_info (read id,
// metadata, basecall summary: everything the batch path needs).  Per-file problems are data: rc[k] (the file stays
// NULL; text as pxg_h5_open would have left it: error[k], 160 bytes each, may be NULL), info[k].status for the read.
extern "C" int pxg_h5_open_many(int64_t n, const char* const* paths, int32_t threads, pxg_h5** files, int32_t* rc,
                                int64_t* n_reads, int32_t* multi, pxg_h5_read_info* first_info, char* error)
{
    if (n < 0 || (n && (!paths || !files || !rc || !n_reads || !multi || !first_info))) return PXG_E_INVALID;
    // A worker batch of single-read files is thousands of files open at once, and a process may hold ~1 000 descriptors
    // (RLIMIT_NOFILE's usual soft limit): beyond a share of the limit a file keeps its MAPPING and gives its descriptor
    // back (copy_out then copies from the map instead of pread: slower for big stretches, never wrong).
    int64_t fd_budget = 256;
    {
        struct rlimit lim;
        if (getrlimit(RLIMIT_NOFILE, &lim) == 0)
            fd_budget = lim.rlim_cur == RLIM_INFINITY ? (int64_t)1 << 40 : std::max<int64_t>((int64_t)lim.rlim_cur / 4, 16);
    }
    std::atomic<int64_t> kept{ 0 };
    run_pool(n, threads, [&](int64_t k) {
        files[k] = nullptr;
        n_reads[k] = 0;
        multi[k] = 0;
        memset(&first_info[k], 0, sizeof(first_info[k]));
        if (error) error[160 * k] = 0;
        rc[k] = paths[k] ? pxg_h5_open_mt(paths[k], 1, &files[k]) : PXG_E_INVALID;
        if (rc[k] != PXG_OK) {
            files[k] = nullptr;
            if (error) {                       // (the message of THIS pool thread's failed open)
                strncpy(error + 160 * k, t_h5_error.c_str(), 159);
                error[160 * k + 159] = 0;
            }
            return;
        }
        n_reads[k] = (int64_t)files[k]->reads.size();
        multi[k] = files[k]->multi ? 1 : 0;
        if (!files[k]->multi && n_reads[k] == 1) (void)pxg_h5_info_mt(files[k], 0, 1, &first_info[k], 1);
        if (kept.fetch_add(1) >= fd_budget && files[k]->fd >= 0) {
            close(files[k]->fd);
            files[k]->fd = -1;
        }
    }, 4);
    return PXG_OK;
}

extern "C" void pxg_h5_close_many(int64_t n, pxg_h5* const* files)
{
    for (int64_t k = 0; files && k < n; k++)
        if (files[k]) pxg_h5_close(files[k]);
}

// sequence + '\n' + quality string, Move / Events `move' column, p_model_state of one read
extern "C" int pxg_h5_basecall(const pxg_h5* h, int64_t i, int64_t text_cap, char* text, int64_t move_cap,
                               uint8_t* move, double* p_model_state_or_null, int32_t* has_pms)
{
    if (!h || i < 0 || i >= (int64_t)h->reads.size()) return PXG_E_INVALID;
    H5_GUARD_BEGIN
    pxg_h5_read_info o;
    memset(&o, 0, sizeof(o));
    std::string fq;
    std::vector<uint8_t> mv;
    std::vector<double> pms;
    basecall_of(h, h->reads[(size_t)i], o, &fq, &mv, &pms);
    if ((int64_t)fq.size() + 1 > text_cap || (int64_t)mv.size() > move_cap) fail(PXG_E_NOMEM, "basecall buffers too small");
    if (text) { memcpy(text, fq.data(), fq.size()); text[fq.size()] = 0; }
    if (move && !mv.empty()) memcpy(move, mv.data(), mv.size());
    if (has_pms) *has_pms = !pms.empty();
    if (p_model_state_or_null && !pms.empty()) memcpy(p_model_state_or_null, pms.data(), pms.size() * sizeof(double));
    return PXG_OK;
    H5_GUARD_END(h)
}

// The columns of a read's BaseCalled_template/Events table the per-read processor uses when the table
// brings its own events (albacore, 14 columns: fast5_file.py:166-181 returns it unchanged; consumers
// signal_analyzer.py:311-326,366-443,183-190): start, length, mean, stdv, move, p_model_state as float64
// (exact for every integer and float width a FAST5 holds below 2^53) and model_state as fixed-width text.
// info = per column (in that order, then model_state): class (0 integer, 1 float, 3 string, -1 absent),
// byte width, signedness -- what a caller needs to rebuild the file's own dtypes.
extern "C" int64_t pxg_h5_events(const pxg_h5* h, int64_t i, int64_t cap_rows, int32_t* info /* 7 x 3 */,
                                 double* start, double* length, double* mean, double* stdv, double* move,
                                 double* p_model_state, char* model_state, int32_t model_state_cap)
{
    if (!h || i < 0 || i >= (int64_t)h->reads.size() || !info) return PXG_E_INVALID;
    H5_GUARD_BEGIN
    const pxg_h5_read& r = h->reads[(size_t)i];
    if (r.analyses_obj == UNDEF) return 0;
    const Object an = h->object(r.analyses_obj);
    std::string best;
    uint64_t best_obj = UNDEF;
    for (const auto& c : h->children(an))
        if (c.first.compare(0, 11, "Basecall_1D") == 0 && (best_obj == UNDEF || c.first > best)) { best = c.first; best_obj = c.second; }
    if (best_obj == UNDEF) return 0;
    const uint64_t ev = h->resolve(best_obj, "BaseCalled_template/Events");
    if (ev == UNDEF) return 0;
    const Dataset d = h->object(ev).ds;
    if (d.type.cls != 6) fail(PXG_E_INVALID, "FAST5: Events is not a table");
    const uint64_t n = d.n_elements();
    h->sane_bytes(n, d.type.size);
    static const char* names[7] = { "start", "length", "mean", "stdv", "move", "p_model_state", "model_state" };
    double* outs[6] = { start, length, mean, stdv, move, p_model_state };
    const Datatype::Member* col[7] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    for (const auto& m : d.type.members)
        for (int k = 0; k < 7; k++)
            if (m.name == names[k]) col[k] = &m;
    for (int k = 0; k < 7; k++) {
        info[3 * k] = col[k] ? col[k]->cls : -1;
        info[3 * k + 1] = col[k] ? (int32_t)col[k]->size : 0;
        info[3 * k + 2] = col[k] ? (int32_t)col[k]->is_signed : 0;
        if (!col[k]) continue;
        if ((uint64_t)col[k]->offset + col[k]->size > d.type.size) fail(PXG_E_INVALID, "FAST5: Events table columns lie outside its rows");
        const bool numeric = k < 6;
        if (numeric && !((col[k]->cls == 0 && (col[k]->size == 1 || col[k]->size == 2 || col[k]->size == 4 || col[k]->size == 8)) ||
                         (col[k]->cls == 1 && (col[k]->size == 4 || col[k]->size == 8))))
            fail(PXG_E_UNSUPPORTED, std::string("FAST5: Events column '") + names[k] + "' has a type this reader does not convert");
        if (!numeric && col[k]->cls != 3) fail(PXG_E_UNSUPPORTED, "FAST5: Events column 'model_state' is not fixed-width text");
    }
    if ((int64_t)n > cap_rows) return (int64_t)n;          // the caller asks again with room for n rows
    if (col[6] && model_state && (int32_t)col[6]->size > model_state_cap) fail(PXG_E_NOMEM, "model_state buffer too narrow");
    std::vector<uint8_t> rows(n * d.type.size);
    h->read_dataset(d, rows.data(), rows.size());
    for (uint64_t q = 0; q < n; q++) {
        const uint8_t* row = rows.data() + q * d.type.size;
        for (int k = 0; k < 6; k++) {
            if (!col[k] || !outs[k]) continue;
            const uint8_t* v = row + col[k]->offset;
            double x = 0.0;
            if (col[k]->cls == 1) {
                if (col[k]->size == 4) { float f; memcpy(&f, v, 4); x = f; } else memcpy(&x, v, 8);
            } else {
                uint64_t u = 0;
                for (int b = (int)col[k]->size - 1; b >= 0; b--) u = (u << 8) | v[b];
                if (col[k]->is_signed) {
                    const int sh = 64 - 8 * (int)col[k]->size;
                    x = (double)((int64_t)(u << sh) >> sh);
                } else x = (double)u;
            }
            outs[k][q] = x;
        }
        if (col[6] && model_state) {
            memset(model_state + q * (size_t)model_state_cap, 0, (size_t)model_state_cap);
            memcpy(model_state + q * (size_t)model_state_cap, row + col[6]->offset, col[6]->size);
        }
    }
    return (int64_t)n;
    H5_GUARD_END(h)
}

// Host threads of the reader: ONE set of worker threads per process, started on first use and
// parked on a condition variable between calls.  A loader call is four short parallel phases
// (read groups, metadata, samples, basecall text) of a few milliseconds each: starting and joining
// 15 threads per phase cost more than the phase's own work on a 16-core host.  One job at a time;
// a caller that finds the workers busy (two loader threads) starts threads of its own as before.
// Nothing may leave fn (the callers catch inside their lambdas).
struct PoolJob {
    void (*call)(void*, int64_t, int64_t);
    void* ctx;
    int64_t n, grain;
    std::atomic<int64_t> next{ 0 };
    void drain()
    {
        for (;;) {
            const int64_t k = next.fetch_add(grain);
            if (k >= n) return;
            call(ctx, k, std::min(n, k + grain));
        }
    }
};

struct Pool {
    std::mutex submit;                  // one job at a time
    std::mutex mu;
    std::condition_variable cv, cv_done;
    PoolJob* job = nullptr;
    uint64_t gen = 0;
    int slots = 0;                      // workers that may still join the current job
    int running = 0;                    // workers inside it
    int n_workers = 0;

    void worker()
    {
        std::unique_lock<std::mutex> lk(mu);
        uint64_t seen = 0;
        for (;;) {
            cv.wait(lk, [&] { return job && gen != seen && slots > 0; });
            seen = gen;
            slots--;
            running++;
            PoolJob* j = job;
            lk.unlock();
            j->drain();
            lk.lock();
            if (--running == 0) cv_done.notify_all();
        }
    }

    // true = the job ran here (on the caller and up to threads - 1 workers)
    bool run(PoolJob& j, int threads)
    {
        std::unique_lock<std::mutex> one(submit, std::try_to_lock);
        if (!one.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(mu);
            while (n_workers < threads - 1) {
                try { std::thread(&Pool::worker, this).detach(); }
                catch (const std::exception&) { break; }        // no more threads to be had: the ones we got
                n_workers++;
            }
            job = &j;
            gen++;
            slots = std::min(threads - 1, n_workers);
            running = 0;
        }
        cv.notify_all();
        j.drain();
        std::unique_lock<std::mutex> lk(mu);
        slots = 0;                      // nobody joins a job whose queue is empty
        cv_done.wait(lk, [&] { return running == 0; });
        job = nullptr;
        return true;
    }
};

// (never destroyed: its threads outlive main(); a forked child starts with a pool of its own,
//  the parent's worker threads do not exist there)
static std::atomic<Pool*> g_pool{ nullptr };
static void pool_after_fork() { g_pool.store(nullptr); }
static Pool* the_pool()
{
    Pool* p = g_pool.load();
    if (p) return p;
    static std::mutex mk;
    std::lock_guard<std::mutex> lk(mk);
    p = g_pool.load();
    if (!p) {
        static bool hooked = false;
        if (!hooked) { pthread_atfork(nullptr, nullptr, pool_after_fork); hooked = true; }
        p = new Pool();
        g_pool.store(p);
    }
    return p;
}

static const int64_t kSmallJob = 512;

// fn(k) for k in [0, n) on `threads` host threads, `grain` consecutive k per queue access
template <typename Fn>
static void run_pool(int64_t n, int threads, Fn fn, int64_t grain)
{
    if (n <= 0) return;
    grain = std::max<int64_t>(1, grain);
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, (n + grain - 1) / grain));
    PoolJob j;
    j.call = [](void* c, int64_t a, int64_t b) { Fn& f = *(Fn*)c; for (int64_t k = a; k < b; k++) f(k); };
    j.ctx = &fn;
    j.n = n;
    j.grain = grain;
    if (nt == 1) { j.drain(); return; }
    if (!getenv("PXG_H5_SPAWN_THREADS") && the_pool()->run(j, nt)) return;
    // the workers are busy with another caller's job.  A small job (a reference-sized worker call: 128 reads, one of
    // dozens in flight on other threads) runs on its caller alone -- the parallelism is between the calls, and
    // starting threads of its own would cost more than its share of the work; a big one starts them as before
    if (n <= kSmallJob && !getenv("PXG_H5_SPAWN_THREADS")) { j.drain(); return; }
    std::vector<std::thread> own;
    own.reserve((size_t)nt);
    for (int t = 1; t < nt; t++) {
        try { own.emplace_back([&j] { j.drain(); }); }
        catch (const std::exception&) { break; }    // no more threads to be had: the ones we got, and this one
    }
    j.drain();
    for (auto& t : own) t.join();
}

// The int16 samples of many reads (any mix of open files), decoded on `threads` host threads
// straight into `arena` (the caller's staging buffer): read k = (files[k], index[k]) goes to
// arena[dst_start[k] .. dst_start[k] + n_samples[k]).  status[k] = 0 or that read's own error code.
extern "C" int pxg_h5_load_signals(int64_t n, const pxg_h5* const* files, const int64_t* index,
                                   const int64_t* dst_start, const int64_t* n_samples, int16_t* arena,
                                   int32_t threads, int32_t* status)
{
    if (n < 0 || (n && (!files || !index || !dst_start || !n_samples || !arena || !status))) return PXG_E_INVALID;
    run_pool(n, threads, [&](int64_t k) {
        status[k] = PXG_OK;
        try {
            const pxg_h5* h = files[k];
            if (!h || index[k] < 0 || index[k] >= (int64_t)h->reads.size()) fail(PXG_E_INVALID, "bad read index");
            const Object sig = h->object(h->reads[(size_t)index[k]].signal_obj);
            if (!sig.has_dataset || sig.ds.type.size != 2 || (int64_t)sig.ds.n_elements() != n_samples[k])
                fail(PXG_E_INVALID, "Signal length differs from the batch layout");
            h->read_dataset(sig.ds, (uint8_t*)(arena + dst_start[k]), (uint64_t)n_samples[k] * 2);
        } catch (const H5Error& e) {
            status[k] = e.code;
        } catch (const std::exception&) {
            status[k] = PXG_E_NOMEM;
        }
    });
    return PXG_OK;
}

// Basecall text and move tables of many reads into columnar arenas (the layout of a read
// bundle's seq_arena / qual_arena / move_arena): read k's sequence and quality string go to
// [seq_start[k], seq_start[k] + seq_len[k]) of the two text arenas, its moves to
// [move_start[k], move_start[k] + n_moves[k]).  Lengths come from pxg_h5_info (bc_seq_len,
// bc_n_moves; reads without a basecall / table have length 0 and are skipped).
extern "C" int pxg_h5_basecall_many(int64_t n, const pxg_h5* const* files, const int64_t* index,
                                    const int64_t* seq_start, const int64_t* seq_len, uint8_t* seq_arena,
                                    uint8_t* qual_arena, const int64_t* move_start, const int64_t* n_moves,
                                    uint8_t* move_arena, int32_t threads, int32_t* status)
{
    if (n < 0 || (n && (!files || !index || !seq_start || !seq_len || !move_start || !n_moves || !status)))
        return PXG_E_INVALID;
    // (a few KB of text per read: a reference-sized call's worth is cheaper copied than handed to other threads)
    run_pool(n, n <= 256 ? 1 : threads, [&](int64_t k) {
        status[k] = PXG_OK;
        if (seq_len[k] <= 0 && n_moves[k] <= 0) return;
        try {
            const pxg_h5* h = files[k];
            if (!h || index[k] < 0 || index[k] >= (int64_t)h->reads.size()) fail(PXG_E_INVALID, "bad read index");
            if (h->bc_where) {
                const BcWhere& w = h->bc_where[(size_t)index[k]];
                if (w.state.load(std::memory_order_acquire) == 1 && (int64_t)w.len == seq_len[k] &&
                    (n_moves[k] <= 0 || (w.move_at != UNDEF && (int64_t)w.n_moves == n_moves[k]))) {
                    if (seq_len[k] > 0) {
                        memcpy(seq_arena + seq_start[k], h->at(w.seq_at, w.len), (size_t)w.len);
                        memcpy(qual_arena + seq_start[k], h->at(w.qual_at, w.len), (size_t)w.len);
                    }
                    if (n_moves[k] > 0) memcpy(move_arena + move_start[k], h->at(w.move_at, w.n_moves), (size_t)w.n_moves);
                    return;
                }
            }
            pxg_h5_read_info o;
            memset(&o, 0, sizeof(o));
            std::string fq;
            std::vector<uint8_t> mv;
            basecall_of(h, h->reads[(size_t)index[k]], o, &fq, &mv, nullptr);
            const size_t nl = fq.find('\n');
            if (nl == std::string::npos || (int64_t)nl != seq_len[k] || (int64_t)(fq.size() - nl - 1) != seq_len[k] ||
                (n_moves[k] > 0 && (int64_t)mv.size() != n_moves[k]))
                fail(PXG_E_INVALID, "basecall of a read differs from the batch layout");
            if (seq_len[k] > 0) {
                memcpy(seq_arena + seq_start[k], fq.data(), (size_t)seq_len[k]);
                memcpy(qual_arena + seq_start[k], fq.data() + nl + 1, (size_t)seq_len[k]);
            }
            if (n_moves[k] > 0) memcpy(move_arena + move_start[k], mv.data(), (size_t)n_moves[k]);
        } catch (const H5Error& e) {
            status[k] = e.code;
        } catch (const std::exception&) {
            status[k] = PXG_E_NOMEM;
        }
    });
    return PXG_OK;
}
