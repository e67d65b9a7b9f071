// k_signal.hip -- HBM-bound per-read signal kernels (gfx950):
//   K1  head pool            a1+a2  fast5_file.py:122-131, signal_loader.py:212-231
//       pool+scale           a1+a5  signal_loader.py:233-264      (test hook)
//       scaler transform+QC  a4     signal_loader.py:98-109
//   K4  barcode window       a9-a11 signal_analyzer.py:445-448, barcoding.py:77-101
//       finalize             a12 tail + a13  barcoding.py:72-75,108-118
// All byte/float work here is bit-exact against the oracle: float64 pA
// conversion, NumPy's 15-element pairwise float32 sum, two-rounding Horner.
#include "pxg_common.h"

// ---------------------------------------------------------------------------
__global__ void k_raw_to_pa(int64_t n, const int16_t* __restrict__ raw, pxg_calib cal,
                            float* __restrict__ out)
{
    const double k = cal.range / cal.digitisation;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = pxg_raw2pa(raw[i], k, cal.offset);
}

int pxg_launch_raw_to_pa(pxg_ctx* ctx, int64_t n, const int16_t* raw, const pxg_calib* cal,
                         float* out)
{
    if (n <= 0) return PXG_OK;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_raw_to_pa, dim3(blocks), dim3(256), 0, ctx->stream, n, raw, *cal, out);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// start of a run: per-read state back to "nothing decided" in ONE launch (it was five
// hipMemsetAsync fills, ~4.4 us each plus the gaps between them)
// ---------------------------------------------------------------------------
__global__ void k_reset_batch(int64_t n_reads, int32_t* __restrict__ counters, int32_t* __restrict__ status,
                              float* __restrict__ pred, float* __restrict__ ss, int32_t* __restrict__ segs)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r < 8) counters[r] = 0;
    if (r >= n_reads) return;
    status[r] = 0;
    reinterpret_cast<float2*>(pred)[r] = make_float2(0.0f, 0.0f);
    reinterpret_cast<float2*>(ss)[r] = make_float2(0.0f, 0.0f);
    int4* sg = reinterpret_cast<int4*>(segs + r * 2 * PXG_N_SEGMENTS);
#pragma unroll
    for (int q = 0; q < 2 * PXG_N_SEGMENTS / 4; q++) sg[q] = make_int4(-1, -1, -1, -1);
}

int pxg_launch_reset_batch(pxg_ctx* ctx, int64_t n)
{
    const int64_t threads = n > 8 ? n : 8;
    hipLaunchKernelGGL(k_reset_batch, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       ctx->counters.p, ctx->status.p, ctx->pred.p, ctx->ss.p, ctx->segs.p);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// K0: samples that crossed PCIe as zig-zag delta bytes (pxg_batch_stage_z, include/pxg.h) back to
// int16.  One WAVE per 1 024-sample chunk, sixteen samples per lane, four chunks per workgroup and
// no workgroup barrier: the byte offset of a lane's samples is a wave prefix sum of (1 + control
// bit), their values a wave prefix sum of the deltas, each as a serial part inside the lane + one
// shuffle scan.  The chunk's bytes are staged through LDS with coalesced dword loads.  (First
// version: a workgroup per chunk, four samples per thread, two block scans -- ten barriers per
// chunk and 1.24 ms for a 10 000-read batch; this one 0.55 ms = 3.5 TB/s of reads + writes.)
// ---------------------------------------------------------------------------
#define ZD_STAGE_DW ((PXG_Z_CTRL_BYTES + 2 * PXG_Z_CHUNK) / 4 + 2)      // + 1 below: the funnel shift's second word

__device__ __forceinline__ int wave_exclusive_scan(int v, int lane, int& total)
{
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    total = __shfl(inc, 63);
    return inc - v;
}

__global__ __launch_bounds__(256) void k_z_decode(int64_t n_chunks, const uint8_t* __restrict__ z, int64_t z_bytes,
                                                  const pxg_z_chunk* __restrict__ chunks, int64_t data_base,
                                                  int64_t dst_base, int16_t* __restrict__ out,
                                                  const int64_t* __restrict__ off, int64_t n_reads, int64_t prefix_limit)
{
    __shared__ unsigned stage_all[4][ZD_STAGE_DW + 2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t g = blockIdx.x * 4ll + wv;
    if (g >= n_chunks) return;                              // whole waves leave; nothing below is block-wide
    const pxg_z_chunk c = chunks[g];
    if (prefix_limit > 0) {
        // pxg_batch_stage_z_prefix: only the chunks that hold the first `prefix_limit` samples of their read
        // crossed the link -- the read is the last one that starts at or before the chunk
        const int64_t at = c.dst - dst_base;
        int64_t lo = 0, hi = n_reads;              // off[lo] <= at < off[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= at) lo = mid; else hi = mid;
        }
        if (at - off[lo] >= prefix_limit) return;
    }
    unsigned* stage = stage_all[wv];
    const uint8_t* src = z + (c.data_off - data_base);
    const int mis = (int)((uintptr_t)src & 3);
    const unsigned* src4 = reinterpret_cast<const unsigned*>(src - mis);
    // never past the stream's buffer (z_bytes + 16 allocated): a chunk near the end that needs fewer bytes
    // than its worst case must not read beyond it (the words not staged are never addressed by a
    // well-formed chunk; pxg_z_check vouches for the rest)
    const int64_t room64 = (z_bytes + 12 - ((c.data_off - data_base) - mis)) >> 2;
    const int room = room64 < 0 ? 0 : (room64 > ZD_STAGE_DW ? ZD_STAGE_DW : (int)room64);
    const int i0 = 16 * lane;                               // samples i0 .. i0 + 15 of the chunk
    const int n_valid = c.len - i0 < 0 ? 0 : (c.len - i0 > 16 ? 16 : c.len - i0);
    int d[16], sum = 0, total;
    if (c.codec == PXG_Z_PACKED) {
        // ---- bit-packed groups of four: header first (it says how many bytes follow), then the bits ----
        int n_dw = min((mis + PXG_Z_CTRL_BYTES + 3) >> 2, room);
        if (lane < n_dw) stage[lane] = src4[lane];          // <= 33 dwords
        __builtin_amdgcn_wave_barrier();
        const uint8_t* hdr = reinterpret_cast<const uint8_t*>(stage) + mis;
        const unsigned codes = (unsigned)hdr[2 * lane] | ((unsigned)hdr[2 * lane + 1] << 8);   // this lane's 4 groups
        int w[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int code = (int)((codes >> (4 * j)) & 15u);
            w[j] = (i0 + 4 * j < c.len) ? (code == 15 ? 16 : code) : 0;     // (a group past the end carries nothing)
            mine += 4 * w[j];
        }
        int bit = wave_exclusive_scan(mine, lane, total) + 8 * (mis + PXG_Z_CTRL_BYTES);
        __builtin_amdgcn_wave_barrier();                    // every lane has read its header bytes
        n_dw = min((mis + PXG_Z_CTRL_BYTES + ((total + 7) >> 3) + 3) >> 2, room);
        for (int k = lane; k < n_dw; k += 64) stage[k] = src4[k];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int wq = w[q >> 2];
            const int at = bit >> 5;                        // (<= 128 + 2 048 bytes + 3: inside the stage buffer)
            const unsigned v = __funnelshift_r(stage[at], stage[at + 1], (unsigned)(bit & 31));
            const unsigned zz = v & ((1u << wq) - 1u);
            bit += wq;
            sum += (int)((zz >> 1) ^ (0u - (zz & 1u)));
            d[q] = sum;
        }
    } else {
    // at most 128 + 2 (len - 1) bytes; the last dword may reach into the next chunk (or the 16
    // spare bytes behind the stream)
    const int n_dw = min((mis + PXG_Z_CTRL_BYTES + 2 * (c.len - 1) + 3) >> 2, room);
    for (int k = lane; k < n_dw; k += 64) stage[k] = src4[k];   // (LDS words not staged are only read for malformed chunks)
    __builtin_amdgcn_wave_barrier();
    const uint8_t* ctrl = reinterpret_cast<const uint8_t*>(stage) + mis;
    const uint8_t* data = ctrl + PXG_Z_CTRL_BYTES;
    unsigned bits = (unsigned)ctrl[2 * lane] | ((unsigned)ctrl[2 * lane + 1] << 8);
    bits &= (1u << n_valid) - 1u;                           // (sample 0's bit is never set)
    const int mine = n_valid + __builtin_popcount(bits) - ((lane == 0 && n_valid > 0) ? 1 : 0);
    int at = wave_exclusive_scan(mine, lane, total);
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const bool has = q < n_valid && (i0 + q) >= 1;      // sample 0 is in the record
        const bool two = (bits >> q) & 1u;
        unsigned zz = 0;
        if (has) zz = data[at];
        if (has && two) zz |= (unsigned)data[at + 1] << 8;
        at += has ? (two ? 2 : 1) : 0;
        sum += (int)((zz >> 1) ^ (0u - (zz & 1u)));         // zig-zag; only the low 16 bits matter
        d[q] = sum;                                          // inclusive, inside the lane
    }
    }
    const int before = wave_exclusive_scan(sum, lane, total) + (int)c.first;
    int16_t* dst = out + (c.dst - dst_base) + i0;
    if (n_valid == 16 && (((uintptr_t)dst) & 3) == 0) {     // eight dword stores
        unsigned* d4 = reinterpret_cast<unsigned*>(dst);
#pragma unroll
        for (int q = 0; q < 8; q++)
            d4[q] = ((unsigned)(before + d[2 * q]) & 0xFFFFu) | ((unsigned)(before + d[2 * q + 1]) << 16);
    } else if (n_valid == 16) {                             // 2-byte aligned: short, seven dwords, short
        dst[0] = (int16_t)(uint16_t)(before + d[0]);
        unsigned* d4 = reinterpret_cast<unsigned*>(dst + 1);
#pragma unroll
        for (int q = 0; q < 7; q++)
            d4[q] = ((unsigned)(before + d[2 * q + 1]) & 0xFFFFu) | ((unsigned)(before + d[2 * q + 2]) << 16);
        dst[15] = (int16_t)(uint16_t)(before + d[15]);
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (q < n_valid) dst[q] = (int16_t)(uint16_t)(before + d[q]);
    }
}

int pxg_launch_z_decode(pxg_ctx* ctx, hipStream_t stream, int64_t n_chunks, const uint8_t* z, int64_t z_bytes,
                        const pxg_z_chunk* chunks, int64_t data_base, int64_t dst_base, int16_t* out, const int64_t* off,
                        int64_t n_reads, int64_t prefix_limit)
{
    if (n_chunks <= 0) return PXG_OK;
    if (n_chunks > 0x7fffffffLL) {
        pxg_set_err(ctx, "pxg_batch_stage_z: too many chunks");
        return PXG_E_INVALID;
    }
    hipLaunchKernelGGL(k_z_decode, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, stream, n_chunks, z, z_bytes, chunks, data_base,
                       dst_base, out, off, n_reads, off ? prefix_limit : 0);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// K1: one thread per output element of the left-padded head (width = 2000).
// ---------------------------------------------------------------------------
__global__ void k_head_pool(int64_t n_reads, const int16_t* __restrict__ raw,
                            const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
                            int length_limit, int stride, int min_length, int width,
                            float* __restrict__ head, int32_t* __restrict__ status)
{
    const int64_t r = blockIdx.x;            // reads on x: gridDim.y stops at 65535
    const int j = blockIdx.y * blockDim.x + threadIdx.x;
    if (r >= n_reads || j >= width) return;
    const int64_t n_raw = off[r + 1] - off[r];
    int64_t L = n_raw < length_limit ? n_raw : length_limit;
    L -= L % stride;
    float* out = head + r * (int64_t)width;
    if (L < min_length) {
        out[j] = 0.0f;
        if (j == 0) status[r] = PXG_ST_SCALER_SIGNAL_TOO_SHORT;
        return;
    }
    if (j == 0) status[r] = PXG_ST_OKAY;
    const int n_means = (int)(L / stride);
    const int pad = width - n_means;
    if (j < pad) {
        out[j] = 0.0f;
        return;
    }
    const pxg_calib c = cal[r];
    const double k = c.range / c.digitisation;
    out[j] = pxg_block_mean(raw + off[r] + (int64_t)(j - pad) * stride, stride, k, c.offset);
}

int pxg_launch_head_pool(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                         const pxg_calib* cal, float* head, int32_t* status)
{
    if (n <= 0) return PXG_OK;
    const int width = ctx->cfg.scaler_length / ctx->cfg.stride;
    dim3 grid((unsigned)n, (width + 255) / 256);
    hipLaunchKernelGGL(k_head_pool, grid, dim3(256), 0, ctx->stream, n, raw, off, cal,
                       ctx->cfg.scaler_length, ctx->cfg.stride, ctx->cfg.scaler_min_length,
                       width, head, status);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
__global__ void k_pool_scale(int64_t n_reads, const int16_t* __restrict__ raw,
                             const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
                             const float* __restrict__ ss, const int64_t* __restrict__ poff,
                             int stride, float* __restrict__ out)
{
    const int64_t r = blockIdx.x;
    if (r >= n_reads) return;
    const int64_t P = (off[r + 1] - off[r]) / stride;
    const pxg_calib c = cal[r];
    const double k = c.range / c.digitisation;
    const float scale = ss[2 * r], shift = ss[2 * r + 1];
    for (int64_t p = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; p < P;
         p += (int64_t)gridDim.y * blockDim.x) {
        float m = pxg_block_mean(raw + off[r] + p * stride, stride, k, c.offset);
        float y = scale * m;
        out[poff[r] + p] = y + shift;
    }
}

int pxg_launch_pool_scale(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                          const pxg_calib* cal, const float* ss, const int64_t* poff,
                          float* out)
{
    if (n <= 0) return PXG_OK;
    dim3 grid((unsigned)n, 16);
    hipLaunchKernelGGL(k_pool_scale, grid, dim3(256), 0, ctx->stream, n, raw, off, cal, ss, poff,
                       ctx->cfg.stride, out);
    return PXG_OK;
}

// The same pooled + scaled signal for ONE stretch per read of the resident batch (the adapter
// stretch of dump_adapter_signal, signal_analyzer.py:450-466): read r contributes the pooled
// positions [first[r], first[r] + (ooff[r+1] - ooff[r])).  A stretch that leaves the read raises
// the flag and writes nothing.
__global__ void k_pooled_stretch(int64_t n_reads, const int16_t* __restrict__ raw,
                                 const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
                                 const float* __restrict__ ss, const int64_t* __restrict__ first,
                                 const int64_t* __restrict__ ooff, int stride, float* __restrict__ out,
                                 int* __restrict__ flag)
{
    const int64_t r = blockIdx.x;
    if (r >= n_reads) return;
    const int64_t len = ooff[r + 1] - ooff[r];
    if (len <= 0) return;
    const int64_t P = (off[r + 1] - off[r]) / stride, p0 = first[r];
    if (p0 < 0 || p0 + len > P) {
        if (threadIdx.x == 0 && blockIdx.y == 0) atomicExch(flag, 1);
        return;
    }
    const pxg_calib c = cal[r];
    const double k = c.range / c.digitisation;
    const float scale = ss[2 * r], shift = ss[2 * r + 1];
    for (int64_t j = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; j < len;
         j += (int64_t)gridDim.y * blockDim.x) {
        float m = pxg_block_mean(raw + off[r] + (p0 + j) * stride, stride, k, c.offset);
        float y = scale * m;
        out[ooff[r] + j] = y + shift;
    }
}

int pxg_launch_pooled_stretch(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                              const pxg_calib* cal, const float* ss, const int64_t* first,
                              const int64_t* ooff, float* out, int* flag)
{
    if (n <= 0) return PXG_OK;
    hipLaunchKernelGGL(k_pooled_stretch, dim3((unsigned)n, 4), dim3(256), 0, ctx->stream, n, raw, off, cal,
                       ss, first, ooff, ctx->cfg.stride, out, flag);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// a4: fl(fl(fl32(std)*p) + fl32(mean)); QC against float32-rounded bounds,
// inclusive.  Rows come from the compacted index list (idx == nullptr: identity).
// ---------------------------------------------------------------------------
struct XfrmParams {
    float s_mean, s_std, h_mean, h_std;
    float qs0, qs1, qh0, qh1;
};

__global__ void k_scaler_transform(int64_t n_rows, const int32_t* __restrict__ idx,
                                   const int32_t* __restrict__ count,
                                   const float* __restrict__ pred, XfrmParams xp,
                                   float* __restrict__ ss, int32_t* __restrict__ status)
{
    const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t lim = count ? (int64_t)*count : n_rows;
    if (row >= n_rows || row >= lim) return;
    const int64_t r = idx ? idx[row] : row;
    float a = xp.s_std * pred[2 * r];
    float scale = a + xp.s_mean;
    float b = xp.h_std * pred[2 * r + 1];
    float shift = b + xp.h_mean;
    ss[2 * r] = scale;
    ss[2 * r + 1] = shift;
    const bool ok = scale >= xp.qs0 && scale <= xp.qs1 && shift >= xp.qh0 && shift <= xp.qh1;
    status[r] = ok ? PXG_ST_OKAY : PXG_ST_SCALING_QC_FAIL;
}

int pxg_launch_scaler_transform(pxg_ctx* ctx, int64_t n, const float* pred, float* ss,
                                int32_t* status, const int32_t* idx, const int32_t* count)
{
    if (n <= 0) return PXG_OK;
    const pxg_config& c = ctx->cfg;
    XfrmParams xp = { (float)c.scaler_xfrm[0], (float)c.scaler_xfrm[1], (float)c.scaler_xfrm[2],
                      (float)c.scaler_xfrm[3], (float)c.scaler_qc_scale[0],
                      (float)c.scaler_qc_scale[1], (float)c.scaler_qc_shift[0],
                      (float)c.scaler_qc_shift[1] };
    hipLaunchKernelGGL(k_scaler_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, n, idx, count, pred, xp, ss, status);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// compaction of reads that go on to the scaler network
// ---------------------------------------------------------------------------
// `ord` (or null): the reads in length order -- a wave keeps 64 neighbours of that order together,
// so the 16-row tiles of K2 hold reads of one length class (and one zero-pad class)
__global__ void k_compact_ok(int64_t n, const int32_t* __restrict__ status, const int32_t* __restrict__ ord,
                             int32_t* __restrict__ idx, int32_t* __restrict__ counter)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t r = i < n ? (ord ? ord[i] : i) : 0;
    const bool keep = i < n && status[r] == PXG_ST_OKAY;
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, 0);
    if (keep) idx[base + __popcll(m & ((1ull << lane) - 1))] = (int32_t)r;
}

int pxg_launch_compact_scaler(pxg_ctx* ctx, int64_t n, const int32_t* status, const int32_t* order, int32_t* idx,
                              int32_t* counter)
{
    if (n <= 0) return PXG_OK;
    hipLaunchKernelGGL(k_compact_ok, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       n, status, order, idx, counter);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// Reads in length order, longest first (a counting sort over 1 024-sample length classes; inside a
// class the order is whatever the atomics make it -- no result depends on it: every read's outputs
// are functions of that read alone).  A sequencing run's lengths spread over two decades
// (signal_analyzer.py:347-349 scans at most segmentation_scan_limit samples of each): blocks of K3
// then hold eight reads of one class instead of waiting for their longest, and K2's tiles share
// their zero-pad prefix (signal_loader.py:227-229).
// ---------------------------------------------------------------------------
#define ORD_CLASSES 1024
__device__ __forceinline__ int order_class(int64_t len)
{
    const int64_t c = len >> 10;
    return ORD_CLASSES - 1 - (int)(c < ORD_CLASSES - 1 ? c : ORD_CLASSES - 1);
}

__global__ __launch_bounds__(1024) void k_order_count(int64_t n, const int64_t* __restrict__ off, int32_t* __restrict__ hist)
{
    __shared__ int h[ORD_CLASSES];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r < n) atomicAdd(&h[order_class(off[r + 1] - off[r])], 1);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// counts -> first slot of every class (one workgroup; a Hillis-Steele scan over 1 024 ints)
__global__ __launch_bounds__(1024) void k_order_starts(int32_t* __restrict__ hist)
{
    __shared__ int a[2][ORD_CLASSES];
    const int t = threadIdx.x;
    const int own = hist[t];
    a[0][t] = own;
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < ORD_CLASSES; d <<= 1) {
        a[cur ^ 1][t] = a[cur][t] + (t >= d ? a[cur][t - d] : 0);
        cur ^= 1;
        __syncthreads();
    }
    hist[t] = a[cur][t] - own;
}

__global__ __launch_bounds__(1024) void k_order_place(int64_t n, const int64_t* __restrict__ off, int32_t* __restrict__ cursor,
                                                       int32_t* __restrict__ order)
{
    __shared__ int h[ORD_CLASSES], base[ORD_CLASSES];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int cls = 0, rank = 0;
    if (r < n) {
        cls = order_class(off[r + 1] - off[r]);
        rank = atomicAdd(&h[cls], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (r < n) order[base[cls] + rank] = (int32_t)r;
}

int pxg_launch_length_order(pxg_ctx* ctx, int64_t n, const int64_t* off, int32_t* order)
{
    if (n <= 0) return PXG_OK;
    int rc = pxg_reserve(ctx, ctx->order_hist, ORD_CLASSES);
    if (rc) return rc;
    PXG_HIP(ctx, hipMemsetAsync(ctx->order_hist.p, 0, ORD_CLASSES * sizeof(int32_t), ctx->stream));
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    hipLaunchKernelGGL(k_order_count, dim3(blocks), dim3(1024), 0, ctx->stream, n, off, ctx->order_hist.p);
    hipLaunchKernelGGL(k_order_starts, dim3(1), dim3(1024), 0, ctx->stream, ctx->order_hist.p);
    hipLaunchKernelGGL(k_order_place, dim3(blocks), dim3(1024), 0, ctx->stream, n, off, ctx->order_hist.p, order);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// K4: barcode window.  One wave per read; the (<= trim) window lives in LDS;
// median by rank counting with a (value, index) total order, which selects the
// same order statistics np.median's partition does.
// ---------------------------------------------------------------------------
#define PXG_MAX_TRIM 512

// Two middle order statistics of v[0..n) (n <= 512) by a most-significant-bit-first radix
// select over order-preserving integer keys: 32 rounds of one ballot + popcount per key slot
// instead of the n^2 comparisons of rank counting (measured: the two medians of the 300-value
// window were ~all of this kernel's 0.29 ms).
__device__ __forceinline__ unsigned median_key(float x)
{
    const unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float median_unkey(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ void wave_median(const float* v, int n, float* sel, int lane)
{
    // sel[0], sel[1] <- the two middle order statistics (equal slots for odd n)
    const int k1 = n / 2, k0 = (n & 1) ? k1 : k1 - 1;
    constexpr int SLOTS = PXG_MAX_TRIM / PXG_WAVE;
    unsigned key[SLOTS];
    bool live[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; q++) {
        const int i = lane + q * PXG_WAVE;
        live[q] = i < n;
        key[q] = live[q] ? median_key(v[i]) : 0xFFFFFFFFu;
    }
    // the k0-th smallest key
    unsigned prefix = 0u, mask = 0u;
    int k = k0;
    for (int bit = 31; bit >= 0; bit--) {
        const unsigned b = 1u << bit;
        int zeros = 0;
#pragma unroll
        for (int q = 0; q < SLOTS; q++)
            zeros += __popcll(__ballot(live[q] && (key[q] & mask) == prefix && !(key[q] & b)));
        if (k >= zeros) {
            k -= zeros;
            prefix |= b;
        }
        mask |= b;
    }
    // the next order statistic: the same key if it occurs often enough, else the smallest larger one
    int n_le = 0;
    unsigned next = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < SLOTS; q++) {
        n_le += __popcll(__ballot(live[q] && key[q] <= prefix));
        if (live[q] && key[q] > prefix && key[q] < next) next = key[q];
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)next, d);
        next = o < next ? o : next;
    }
    if (lane == 0) {
        sel[0] = median_unkey(prefix);
        sel[1] = (k1 == k0 || n_le > k1) ? median_unkey(prefix) : median_unkey(next);
    }
    __syncthreads();
}

__device__ __forceinline__ void window_normalize(float* xs, float* ys, float* sel, int n,
                                                 int trim, float pad_filler,
                                                 float* __restrict__ out, int lane)
{
    wave_median(xs, n, sel, lane);
    float med;
    {
        float s = sel[0] + sel[1];
        med = (n & 1) ? sel[1] : s / 2.0f;
        med = (med == 0.0f) ? 0.0f : med;      // a zero median is +0 (which of several +-0 samples a
                                               // partition leaves in the middle is not defined)
    }
    __syncthreads();
    for (int i = lane; i < n; i += PXG_WAVE) ys[i] = __builtin_fabsf(xs[i] - med);
    __syncthreads();
    wave_median(ys, n, sel, lane);
    float mad;
    {
        float s = sel[0] + sel[1];
        mad = (n & 1) ? sel[1] : s / 2.0f;
    }
    const double dd = fmax(0.01, (double)mad * 1.4826);
    const float div = (float)dd;
    const int pad = trim - n;
    for (int i = lane; i < pad; i += PXG_WAVE) out[i] = pad_filler;
    for (int i = lane; i < n; i += PXG_WAVE) out[pad + i] = (xs[i] - med) / div;
}

__global__ __launch_bounds__(64) void k_barcode_window_raw(
    int64_t n_reads, const int16_t* __restrict__ raw, const int64_t* __restrict__ off,
    const pxg_calib* __restrict__ cal, const float* __restrict__ ss,
    const int32_t* __restrict__ status, const int32_t* __restrict__ segs, int adapter_state,
    int stride, int minlen, int maxlen, int trim, float pad_filler, float* __restrict__ win,
    int32_t* __restrict__ idx_demux, int32_t* __restrict__ counter)
{
    __shared__ float xs[PXG_MAX_TRIM], ys[PXG_MAX_TRIM], sel[2];
    const int64_t r = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[r] != PXG_ST_OKAY) return;
    const int a0 = segs[r * 2 * PXG_N_SEGMENTS + adapter_state];
    const int a1 = segs[r * 2 * PXG_N_SEGMENTS + PXG_N_SEGMENTS + adapter_state];
    if (a0 < 0) return;
    const int len = a1 - a0 + 1;
    if (len <= 0 || len < minlen || len > maxlen) return;
    const int n = len < trim ? len : trim;
    const int start = a1 + 1 - n;
    const pxg_calib c = cal[r];
    const double k = c.range / c.digitisation;
    const float scale = ss[2 * r], shift = ss[2 * r + 1];
    for (int i = lane; i < n; i += PXG_WAVE) {
        float m = pxg_block_mean(raw + off[r] + (int64_t)(start + i) * stride, stride, k, c.offset);
        float y = scale * m;
        xs[i] = y + shift;
    }
    __syncthreads();
    window_normalize(xs, ys, sel, n, trim, pad_filler, win + r * (int64_t)trim, lane);
    if (lane == 0) idx_demux[atomicAdd(counter, 1)] = (int32_t)r;
}

int pxg_launch_barcode_window_raw(pxg_ctx* ctx, int64_t n, const int16_t* raw,
                                  const int64_t* off, const pxg_calib* cal, const float* ss,
                                  const int32_t* status, const int32_t* segs, float* win,
                                  int32_t* idx_demux, int32_t* counter)
{
    if (n <= 0) return PXG_OK;
    const pxg_config& c = ctx->cfg;
    hipLaunchKernelGGL(k_barcode_window_raw, dim3((unsigned)n), dim3(64), 0, ctx->stream, n, raw,
                       off, cal, ss, status, segs, c.segmentation_model.adapter_state, c.stride,
                       c.minimum_dna_length, c.maximum_dna_length, c.signal_trim_length,
                       c.pad_filler, win, idx_demux, counter);
    return PXG_OK;
}

__global__ __launch_bounds__(64) void k_barcode_window_f32(
    int64_t n_reads, const float* __restrict__ sig, const int64_t* __restrict__ off, int minlen,
    int maxlen, int trim, float pad_filler, float* __restrict__ win, int8_t* __restrict__ pushed)
{
    __shared__ float xs[PXG_MAX_TRIM], ys[PXG_MAX_TRIM], sel[2];
    const int64_t r = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t len = off[r + 1] - off[r];
    if (len <= 0 || len < minlen || len > maxlen) {
        if (lane == 0) pushed[r] = 0;
        return;
    }
    const int n = len < trim ? (int)len : trim;
    const float* src = sig + off[r] + (len - n);
    for (int i = lane; i < n; i += PXG_WAVE) xs[i] = src[i];
    __syncthreads();
    window_normalize(xs, ys, sel, n, trim, pad_filler, win + r * (int64_t)trim, lane);
    if (lane == 0) pushed[r] = 1;
}

int pxg_launch_barcode_window_f32(pxg_ctx* ctx, int64_t n, const float* sig,
                                  const int64_t* off, float* win, int8_t* pushed)
{
    if (n <= 0) return PXG_OK;
    const pxg_config& c = ctx->cfg;
    hipLaunchKernelGGL(k_barcode_window_f32, dim3((unsigned)n), dim3(64), 0, ctx->stream, n, sig,
                       off, c.minimum_dna_length, c.maximum_dna_length, c.signal_trim_length,
                       c.pad_filler, win, pushed);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// finalize: assemble the per-read result record (signal_loader.py:165-198
// numeric fields; barcoding.py:72-75,108-118)
// ---------------------------------------------------------------------------
struct FinalizeParams {
    int stride, n_classes, n_decoy, n_calibration, adapter_state, trim;
    double score_threshold;
    uint32_t stage_mask;
};

__global__ void k_finalize(int64_t n, FinalizeParams fp, const int64_t* __restrict__ off,
                           const int32_t* __restrict__ status, const float* __restrict__ ss,
                           const float* __restrict__ pred, const int32_t* __restrict__ segs,
                           const float* __restrict__ probs, const int8_t* __restrict__ pushed,
                           const double* __restrict__ calibration,
                           const int32_t* __restrict__ polya,
                           pxg_read_result* __restrict__ out)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n) return;
    pxg_read_result o;
    memset(&o, 0, sizeof(o));
    o.status = status[r];
    o.n_pooled = (int32_t)((off[r + 1] - off[r]) / fp.stride);
    for (int s = 0; s < PXG_N_SEGMENTS; s++) o.seg_first[s] = o.seg_last[s] = -1;
    o.bc_label = -1;
    const bool scaled = o.status != PXG_ST_SCALER_SIGNAL_TOO_SHORT;
    if (scaled) {
        o.scale = ss[2 * r];
        o.shift = ss[2 * r + 1];
        o.scaler_pred[0] = pred[2 * r];
        o.scaler_pred[1] = pred[2 * r + 1];
    }
    if ((fp.stage_mask & PXG_STAGE_SEGMENT) &&
        (o.status == PXG_ST_OKAY || o.status == PXG_ST_ADAPTER_NOT_DETECTED)) {
        for (int s = 0; s < PXG_N_SEGMENTS; s++) {
            o.seg_first[s] = segs[r * 2 * PXG_N_SEGMENTS + s];
            o.seg_last[s] = segs[r * 2 * PXG_N_SEGMENTS + PXG_N_SEGMENTS + s];
        }
    }
    if ((fp.stage_mask & PXG_STAGE_BARCODE) && o.status == PXG_ST_OKAY && pushed[r]) {
        o.bc_pushed = 1;
        const float* p = probs + r * PXG_MAX_CLASSES;
        int arg = 0;
        for (int j = 1; j < fp.n_classes; j++)
            if (p[j] > p[arg]) arg = j;
        for (int j = 0; j < fp.n_classes; j++) o.probs[j] = p[j];
        const int label = arg - fp.n_decoy;
        const float score = p[arg];
        o.bc_label = (int8_t)label;
        o.bc_score = score;
        o.bc_called = (label >= 0 && (double)score >= fp.score_threshold) ? 1 : 0;
        int ph = 0;
        if (!(score <= 0.0f)) {            // bisect_right over the float64 table
            const double x = (double)score;
            int lo = 0, hi = fp.n_calibration;
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                if (x < calibration[mid]) hi = mid; else lo = mid + 1;
            }
            ph = lo;
        }
        o.bc_phred = (uint8_t)ph;
    }
    if ((fp.stage_mask & PXG_STAGE_POLYA) && polya && o.status == PXG_ST_OKAY) {
        const int32_t* po = polya + r * 8;
        o.polya_called = (int8_t)po[0];
        o.polya_n_spikes = po[1];
        o.polya_dwell_samples = po[2];
        o.polya_begin = (int64_t)(((uint64_t)(uint32_t)po[4] << 32) | (uint32_t)po[3]);
        o.polya_end = (int64_t)(((uint64_t)(uint32_t)po[6] << 32) | (uint32_t)po[5]);
    }
    out[r] = o;
}

__global__ void k_mark_pushed(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                              int8_t* __restrict__ pushed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *count) pushed[idx[i]] = 1;
}

int pxg_launch_finalize(pxg_ctx* ctx, int64_t n, uint32_t stage_mask)
{
    if (n <= 0) return PXG_OK;
    const pxg_config& c = ctx->cfg;
    // pushed flags live in the first n bytes of the (reused) bidir scratch? no:
    // keep them in the tail of the status arena (int8 view of n extra int32s)
    int8_t* pushed = (int8_t*)(ctx->status.p + n);
    (void)hipMemsetAsync(pushed, 0, (size_t)n, ctx->stream);
    if (stage_mask & PXG_STAGE_BARCODE)
        hipLaunchKernelGGL(k_mark_pushed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           ctx->stream, ctx->idx_demux.p, ctx->counters.p + 1, pushed);
    FinalizeParams fp = { c.stride, c.demux_dense.out_dim, c.number_of_decoy_labels,
                          c.n_calibration, c.segmentation_model.adapter_state,
                          c.signal_trim_length, c.score_threshold, stage_mask };
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, n,
                       fp, ctx->offsets.p, ctx->status.p, ctx->ss.p, ctx->pred.p, ctx->segs.p,
                       ctx->probs.p, pushed, ctx->d_calibration,
                       (stage_mask & PXG_STAGE_POLYA) ? ctx->polya_out.p : nullptr, ctx->results.p);
    return PXG_OK;
}
