// pxg_api.hip -- C ABI of libpxg.so (include/pxg.h): context lifetime, batch
// residency, stage orchestration on one HIP stream, per-stage HIP-event timers.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include "pxg_common.h"
#include <cctype>
#include "pxg_zcheck.h"

static std::string g_last_error;

static int fail(pxg_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) pxg_set_err(ctx, msg);
    else g_last_error = msg;
    return code;
}

void pxg_timer_begin(pxg_ctx* ctx, int t)
{
    (void)hipEventRecord(ctx->ev_start[t], ctx->stream);
    ctx->ev_used[t] = true;
    ctx->launches[t]++;
}

void pxg_timer_end(pxg_ctx* ctx, int t)
{
    (void)hipEventRecord(ctx->ev_stop[t], ctx->stream);
}

// ---------------------------------------------------------------------------
// HMM tables (worker_persistence.py:95-121 -> device image).  All logs are
// taken on the host with the same libm the oracle uses.
// ---------------------------------------------------------------------------
#define PXG_SQRT_2_PI 2.50662827463   /* pomegranate's constant */

static int build_hmm(pxg_ctx* ctx, const pxg_hmm& h, PxgHmmDev& d)
{
    memset(&d, 0, sizeof(d));
    if (h.n_states < 1 || h.n_states > PXG_MAX_STATES)
        return fail(ctx, PXG_E_INVALID, "HMM: bad state count");
    d.n_states = h.n_states;
    d.adapter_state = h.adapter_state;
    d.polya_state = h.polya_state;
    const int S = h.n_states;
    for (int s = 0; s < S; s++) {
        if (h.name_rank[s] < 0 || h.name_rank[s] >= S)
            return fail(ctx, PXG_E_INVALID, "HMM: bad name_rank");
        d.order[h.name_rank[s]] = s;
    }
    d.left_to_right = 1;
    d.max_in = 0;
    for (int s = 0; s < S; s++) {
        d.log_start[s] = h.start_prob[s] > 0.0 ? log(h.start_prob[s]) : -INFINITY;
        int nin = 0;
        for (int r = 0; r < S; r++) {               // sources in name-sorted order
            const int k = d.order[r];
            if (h.trans[k][s] > 0.0) {
                d.in_src[s][nin] = k;
                d.in_logp[s][nin] = log(h.trans[k][s]);
                nin++;
                if (k > s) d.left_to_right = 0;
                else d.shift_mask |= 1u << (s - k);
            }
        }
        for (int q = nin; q < PXG_MAX_STATES; q++) {
            d.in_src[s][q] = -1;
            d.in_logp[s][q] = -INFINITY;
        }
        d.max_in = std::max(d.max_in, nin);
        const int nm = h.n_mix[s];
        if (nm < 1 || nm > PXG_MAX_MIXTURE) return fail(ctx, PXG_E_INVALID, "HMM: bad mixture");
        d.n_mix[s] = nm;
        double wsum = 0.0;
        for (int k = 0; k < nm; k++) wsum += h.mix_weight[s][k];
        for (int k = 0; k < nm; k++) {
            const double sd = h.mix_sigma[s][k];
            d.mu[s][k] = h.mix_mu[s][k];
            d.lssp[s][k] = -log(sd * PXG_SQRT_2_PI);
            d.tss[s][k] = 1.0 / (2.0 * sd * sd);
            d.logw[s][k] = nm > 1 ? log(h.mix_weight[s][k] / wsum) : 0.0;
        }
    }
    return PXG_OK;
}

// Sigmoid spline table: float64 construction from +,*,fma only (no libm), so
// the oracle and this library hold bit-identical coefficients on any host.
static double det_exp(double x)
{
    const double magic = 6755399441055744.0;        /* 1.5 * 2^52 */
    double kf = (x * 1.4426950408889634 + magic) - magic;
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double fact = 6227020800.0;                      /* 13! */
    double p = 1.0 / fact;
    for (int n = 13; n >= 1; n--) {
        fact /= (double)n;
        p = fma(p, r, 1.0 / fact);
    }
    int64_t bits;
    memcpy(&bits, &p, 8);
    bits += (int64_t)kf << 52;
    memcpy(&p, &bits, 8);
    return p;
}

static void build_sigmoid_table(std::vector<float>& tab)
{
    tab.resize((size_t)PXG_SIG_NSEG * 4);
    const double h = 1.0 / 16.0;
    for (int i = 0; i < PXG_SIG_NSEG; i++) {
        const double z0 = (double)(i - PXG_SIG_HALF) * h, z1 = z0 + h;
        const double s0 = 1.0 / (1.0 + det_exp(-z0)), s1 = 1.0 / (1.0 + det_exp(-z1));
        const double d0 = s0 * (1.0 - s0), d1 = s1 * (1.0 - s1);
        tab[4 * i + 0] = (float)s0;
        tab[4 * i + 1] = (float)(h * d0);
        tab[4 * i + 2] = (float)(3.0 * (s1 - s0) - h * (2.0 * d0 + d1));
        tab[4 * i + 3] = (float)(2.0 * (s0 - s1) + h * (d0 + d1));
    }
}

template <typename T>
static int upload(pxg_ctx* ctx, T** dst, const T* src, size_t n)
{
    if (!src || !n) return fail(ctx, PXG_E_INVALID, "missing weight array");
    PXG_HIP(ctx, hipMalloc((void**)dst, n * sizeof(T)));
    PXG_HIP(ctx, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PXG_OK;
}

// LSTM weights go to the device with their gate columns pre-scaled by the
// activation-table step (Keras column order i, f, g, o: x16, x16, x32, x16);
// exact (powers of two), see k_lstm.hip cell_update.
static int upload_lstm(pxg_ctx* ctx, const pxg_lstm_layer& h, PxgLstmDev& d)
{
    d.input_dim = h.input_dim;
    d.units = h.units;
    if (h.input_dim < 1 || h.units < 1) return fail(ctx, PXG_E_INVALID, "bad LSTM dims");
    if (!h.kernel || !h.recurrent || !h.bias) return fail(ctx, PXG_E_INVALID, "missing weight array");
    const size_t H = (size_t)h.units;
    auto scaled = [&](const float* src, size_t rows) {
        std::vector<float> v(rows * 4 * H);
        for (size_t r = 0; r < rows; r++)
            for (size_t c = 0; c < 4 * H; c++)
                v[r * 4 * H + c] = src[r * 4 * H + c] * ((c / H) == 2 ? 32.0f : 16.0f);
        return v;
    };
    int rc;
    const std::vector<float> k = scaled(h.kernel, (size_t)h.input_dim);
    const std::vector<float> u = scaled(h.recurrent, H);
    const std::vector<float> b = scaled(h.bias, 1);
    if ((rc = upload(ctx, &d.kernel, k.data(), k.size()))) return rc;
    if ((rc = upload(ctx, &d.recurrent, u.data(), u.size()))) return rc;
    return upload(ctx, &d.bias, b.data(), b.size());
}

static int upload_dense(pxg_ctx* ctx, const pxg_dense_layer& h, PxgDenseDev& d)
{
    d.in_dim = h.in_dim;
    d.out_dim = h.out_dim;
    int rc;
    if ((rc = upload(ctx, &d.kernel, h.kernel, (size_t)h.in_dim * h.out_dim))) return rc;
    return upload(ctx, &d.bias, h.bias, (size_t)h.out_dim);
}

extern "C" int pxg_abi_version(void) { return PXG_ABI_VERSION; }

extern "C" const char* pxg_last_error(const pxg_ctx* ctx)
{
    if (!ctx) return g_last_error.c_str();
    static thread_local std::string mine;       // a copy: another thread may fail while this one reads
    {
        std::lock_guard<std::mutex> g(const_cast<pxg_ctx*>(ctx)->mt_err);
        mine = ctx->err;
    }
    return mine.c_str();
}

extern "C" int pxg_create(const pxg_config* cfg, pxg_ctx** out)
{
    if (!cfg || !out) return fail(nullptr, PXG_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != PXG_ABI_VERSION)
        return fail(nullptr, PXG_E_INVALID, "pxg_config.abi_version mismatch");
    if (cfg->stride < 1 || cfg->stride > 128 || cfg->signal_trim_length < 1 ||
        cfg->n_calibration < 0 || cfg->n_calibration > PXG_MAX_CALIBRATION)
        return fail(nullptr, PXG_E_INVALID, "config out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, PXG_E_NODEVICE, "no HIP device visible (this library has no CPU path)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(nullptr, PXG_E_NODEVICE, "device_id out of range");
    pxg_ctx* ctx = new pxg_ctx();
    ctx->cfg = *cfg;
    ctx->device = cfg->device_id;
    memset(ctx->ev_used, 0, sizeof(ctx->ev_used));
    memset(ctx->launches, 0, sizeof(ctx->launches));
    int rc = PXG_OK;
    do {
        if (hipSetDevice(ctx->device) != hipSuccess) { rc = fail(ctx, PXG_E_HIP, "hipSetDevice"); break; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) { rc = fail(ctx, PXG_E_HIP, "props"); break; }
        ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&ctx->scan_stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&ctx->polya_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_polya_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_segmented, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_scan_gate, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_scan_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_staged, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_run_done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_run_done[1], hipEventDisableTiming) != hipSuccess) {
            rc = fail(ctx, PXG_E_HIP, "hipStreamCreate"); break;
        }
        for (int t = 0; t < PXG_N_TIMERS; t++) {
            (void)hipEventCreate(&ctx->ev_start[t]);
            (void)hipEventCreate(&ctx->ev_stop[t]);
        }
        if ((rc = build_hmm(ctx, cfg->segmentation_model, ctx->hmm[0]))) break;
        if ((rc = build_hmm(ctx, cfg->unsplit_model, ctx->hmm[1]))) break;
        if ((rc = upload_lstm(ctx, cfg->scaler_lstm1, ctx->scaler1))) break;
        if ((rc = upload_lstm(ctx, cfg->scaler_lstm2, ctx->scaler2))) break;
        if ((rc = upload_dense(ctx, cfg->scaler_dense, ctx->scaler_dense))) break;
        if ((rc = upload_lstm(ctx, cfg->demux_fwd, ctx->demux_fwd))) break;
        if ((rc = upload_lstm(ctx, cfg->demux_bwd, ctx->demux_bwd))) break;
        if ((rc = upload_lstm(ctx, cfg->demux_top, ctx->demux_top))) break;
        if ((rc = upload_dense(ctx, cfg->demux_dense, ctx->demux_dense))) break;
        if ((rc = upload(ctx, &ctx->d_calibration, cfg->calibration, PXG_MAX_CALIBRATION))) break;
        {
            std::vector<float> tab;
            build_sigmoid_table(tab);
            if ((rc = upload(ctx, &ctx->d_sigtab, tab.data(), tab.size()))) break;
        }
        {   // tables of pxg_log1pexp (pxg_common.h), from the host's libm
            std::vector<double> lt(PXG_LSE_TAB_DOUBLES);
            for (int j = 0; j < 64; j++) lt[j] = exp2((double)j / 64.0);
            for (int j = 0; j <= 128; j++) {
                const double c = 1.0 + (double)j / 128.0;
                lt[64 + j] = log(c);
                lt[64 + 129 + j] = 1.0 / c;
            }
            if ((rc = upload(ctx, &ctx->d_lsetab, lt.data(), lt.size()))) break;
        }
        if ((rc = pxg_lstm_upload(ctx))) break;
        if (cfg->lstm_arith != PXG_LSTM_Q8 && cfg->lstm_arith != PXG_LSTM_F32) {
            rc = fail(ctx, PXG_E_INVALID, "pxg_config.lstm_arith: unknown arithmetic"); break;
        }
        if ((rc = pxg_q8_upload(ctx))) break;
        // measurement knobs (results never depend on them; tests/test_gpu_parity.py checks that)
        ctx->length_order = getenv("PXG_NO_LENGTH_ORDER") == nullptr;
        ctx->prefix_skip = getenv("PXG_NO_PREFIX_SKIP") == nullptr;
        ctx->scan_overlap = getenv("PXG_NO_SCAN_OVERLAP") == nullptr;
        ctx->polya_overlap = getenv("PXG_NO_POLYA_OVERLAP") == nullptr;
        ctx->merge_small_calls = getenv("PXG_NO_CALL_MERGE") == nullptr;
        if (cfg->lstm_arith == PXG_LSTM_Q8 && (rc = pxg_q8_scaler_trajectory(ctx))) break;
    } while (0);
    // host pointers in the copied config are not retained
    ctx->cfg.scaler_lstm1.kernel = ctx->cfg.scaler_lstm1.recurrent = ctx->cfg.scaler_lstm1.bias = nullptr;
    ctx->cfg.scaler_lstm2.kernel = ctx->cfg.scaler_lstm2.recurrent = ctx->cfg.scaler_lstm2.bias = nullptr;
    ctx->cfg.demux_fwd.kernel = ctx->cfg.demux_fwd.recurrent = ctx->cfg.demux_fwd.bias = nullptr;
    ctx->cfg.demux_bwd.kernel = ctx->cfg.demux_bwd.recurrent = ctx->cfg.demux_bwd.bias = nullptr;
    ctx->cfg.demux_top.kernel = ctx->cfg.demux_top.recurrent = ctx->cfg.demux_top.bias = nullptr;
    ctx->cfg.scaler_dense.kernel = ctx->cfg.scaler_dense.bias = nullptr;
    ctx->cfg.demux_dense.kernel = ctx->cfg.demux_dense.bias = nullptr;
    if (rc != PXG_OK) {
        g_last_error = ctx->err;
        pxg_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return PXG_OK;
}

template <typename T>
static void release(DevBuf<T>& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

static void free_lstm(PxgLstmDev& d)
{
    if (d.kernel) (void)hipFree(d.kernel);
    if (d.recurrent) (void)hipFree(d.recurrent);
    if (d.bias) (void)hipFree(d.bias);
}

extern "C" void pxg_destroy(pxg_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    release(ctx->raw); release(ctx->offsets); release(ctx->calib); release(ctx->inject);
    release(ctx->head); release(ctx->pred); release(ctx->ss); release(ctx->status);
    release(ctx->segs); release(ctx->idx_scaler); release(ctx->idx_demux);
    release(ctx->counters); release(ctx->win); release(ctx->bidir); release(ctx->probs);
    release(ctx->lstm_q); release(ctx->lstm_state); release(ctx->lstm_err);
    release(ctx->demux_q); release(ctx->demux_state);
    release(ctx->spare.raw); release(ctx->spare.offsets); release(ctx->spare.calib); release(ctx->spare.inject);
    release(ctx->results); release(ctx->spare.z); release(ctx->spare.zchunks); release(ctx->unsplit_q); release(ctx->vit_bp); release(ctx->polya_ev); release(ctx->polya_over); release(ctx->polya_retry); release(ctx->polya_order); release(ctx->polya_out); release(ctx->spikes);
    release(ctx->ev_tstart); release(ctx->ev_first); release(ctx->ev_off); release(ctx->ev_mean); release(ctx->ev_scaled);
    release(ctx->unsplit_scr); release(ctx->unsplit_iv); release(ctx->unsplit_cnt); release(ctx->unsplit_ivoff);
    release(ctx->unsplit_cand); release(ctx->unit_off); release(ctx->n_win);
    free_lstm(ctx->scaler1); free_lstm(ctx->scaler2); free_lstm(ctx->demux_fwd);
    free_lstm(ctx->demux_bwd); free_lstm(ctx->demux_top);
    if (ctx->scaler_dense.kernel) (void)hipFree(ctx->scaler_dense.kernel);
    if (ctx->scaler_dense.bias) (void)hipFree(ctx->scaler_dense.bias);
    if (ctx->demux_dense.kernel) (void)hipFree(ctx->demux_dense.kernel);
    if (ctx->demux_dense.bias) (void)hipFree(ctx->demux_dense.bias);
    pxg_q8_free(ctx);
    if (ctx->d_calibration) (void)hipFree(ctx->d_calibration);
    if (ctx->d_sigtab) (void)hipFree(ctx->d_sigtab);
    if (ctx->d_lsetab) (void)hipFree(ctx->d_lsetab);
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    for (int k = 0; k < 16; k++) {
        if (ctx->ev_up[k]) { (void)hipEventSynchronize(ctx->ev_up[k]); (void)hipEventDestroy(ctx->ev_up[k]); }
        if (ctx->h_up[k]) (void)hipHostFree(ctx->h_up[k]);
    }
    for (auto& set : ctx->h_meta)
        for (auto& m : set)
            if (m.p) (void)hipHostFree(m.p);
    if (ctx->stream) {
        for (int t = 0; t < PXG_N_TIMERS; t++) {
            (void)hipEventDestroy(ctx->ev_start[t]);
            (void)hipEventDestroy(ctx->ev_stop[t]);
        }
        (void)hipStreamDestroy(ctx->stream);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->scan_stream) (void)hipStreamDestroy(ctx->scan_stream);
        if (ctx->polya_stream) (void)hipStreamDestroy(ctx->polya_stream);
        if (ctx->ev_polya_done) (void)hipEventDestroy(ctx->ev_polya_done);
        if (ctx->ev_scan_gate) (void)hipEventDestroy(ctx->ev_scan_gate);
        if (ctx->ev_segmented) (void)hipEventDestroy(ctx->ev_segmented);
        if (ctx->ev_scan_done) (void)hipEventDestroy(ctx->ev_scan_done);
        if (ctx->ev_staged) (void)hipEventDestroy(ctx->ev_staged);
        for (int q = 0; q < 2; q++)
            if (ctx->ev_run_done[q]) (void)hipEventDestroy(ctx->ev_run_done[q]);
    }
    delete ctx;
}

extern "C" int pxg_get_device_info(pxg_ctx* ctx, pxg_device_info* out)
{
    if (!ctx || !out) return PXG_E_INVALID;
    hipDeviceProp_t prop;
    PXG_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    memset(out, 0, sizeof(*out));
    strncpy(out->name, prop.name, sizeof(out->name) - 1);
    strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    if (!out->name[0]) {
        // some containers give the HIP runtime no marketing name: ask the PCI device's sysfs node, then say
        // what IS known instead of printing nothing
        char bus[32] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), ctx->device) == hipSuccess) {
            for (char* c = bus; *c; c++) *c = (char)tolower(*c);
            const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/product_name";
            if (FILE* fh = fopen(path.c_str(), "r")) {
                if (fgets(out->name, sizeof(out->name), fh)) {
                    for (char* c = out->name; *c; c++)
                        if (*c == '\n') *c = 0;
                }
                fclose(fh);
            }
        }
        // (sysfs often only says "AMD Radeon Graphics": add what identifies the part)
        char arch[24] = {0};
        strncpy(arch, prop.gcnArchName, sizeof(arch) - 1);
        if (char* colon = strchr(arch, ':')) *colon = 0;
        const std::string base = out->name[0] ? out->name : "AMD GPU";
        snprintf(out->name, sizeof(out->name), "%.60s (%s, %d CUs; no product name from the HIP runtime)", base.c_str(), arch,
                 prop.multiProcessorCount);
    }
    out->compute_units = prop.multiProcessorCount;
    out->wavefront_size = prop.warpSize;
    out->total_mem = (int64_t)prop.totalGlobalMem;
    out->lds_per_cu = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    out->clock_khz = prop.clockRate;
    return PXG_OK;
}

extern "C" int pxg_device_pci_bus_id(int device, char* out, int cap)
{
    if (!out || cap < 16) return PXG_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(nullptr, PXG_E_NODEVICE, "pxg_device_pci_bus_id: no such HIP device");
    if (hipDeviceGetPCIBusId(out, cap, device) != hipSuccess)
        return fail(nullptr, PXG_E_HIP, "hipDeviceGetPCIBusId failed");
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// batch residency
// ---------------------------------------------------------------------------
static int reserve_batch(pxg_ctx* ctx, int64_t n, int64_t n_samples)
{
    const pxg_config& c = ctx->cfg;
    const size_t width = (size_t)(c.scaler_length / c.stride);
    int rc;
    if ((rc = pxg_reserve(ctx, ctx->raw, (size_t)n_samples + 64))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->offsets, (size_t)n + 1))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->calib, (size_t)n))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->inject, (size_t)n * 2))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->head, (size_t)n * width))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->pred, (size_t)n * 2))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->ss, (size_t)n * 2))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->status, (size_t)n * 2 + 16))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->segs, (size_t)n * 2 * PXG_N_SEGMENTS))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->idx_scaler, (size_t)n))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->idx_demux, (size_t)n))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->counters, 8))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->win, (size_t)n * c.signal_trim_length))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->bidir, (size_t)((n + 15) / 16 * 16) * c.signal_trim_length * 96))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->probs, (size_t)n * PXG_MAX_CLASSES))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->results, (size_t)n))) return rc;
    return PXG_OK;
}

// sampling-rate range of a batch: sizes the window scratch of the chimera scan without a
// device round trip (signal_analyzer.py:374-383: windows are int(seconds * rate) samples)
static int64_t longest_of(const int64_t* off, int64_t n)
{
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) m = std::max(m, off[i + 1] - off[i]);
    return m;
}

static void rate_range(const pxg_calib* calib, int64_t n, double& lo, double& hi)
{
    lo = hi = n > 0 ? calib[0].sampling_rate : 0.0;
    for (int64_t i = 1; i < n; i++) {
        lo = std::min(lo, calib[i].sampling_rate);
        hi = std::max(hi, calib[i].sampling_rate);
    }
}

extern "C" int pxg_batch_upload(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                                const int64_t* raw_offsets, const pxg_calib* calib,
                                const float* scale_shift_or_null)
{
    if (!ctx) return PXG_E_INVALID;
    if (n_reads < 0 || (n_reads > 0 && (!raw_offsets || !calib)))
        return fail(ctx, PXG_E_INVALID, "pxg_batch_upload: bad arguments");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    ctx->n_reads = 0;
    ctx->polya_unsettled = false;     // results not downloaded by now are given up
    if (n_reads == 0) return PXG_OK;
    for (int64_t i = 0; i < n_reads; i++)
        if (raw_offsets[i + 1] < raw_offsets[i])
            return fail(ctx, PXG_E_INVALID, "raw_offsets must be non-decreasing");
    if (raw_offsets[0] != 0) return fail(ctx, PXG_E_INVALID, "raw_offsets[0] must be 0");
    const int64_t n_samples = raw_offsets[n_reads];
    if (n_samples > 0 && !raw_arena) return fail(ctx, PXG_E_INVALID, "raw_arena is null");
    if (n_reads > (1LL << 30)) return fail(ctx, PXG_E_INVALID, "too many reads");
    int rc = reserve_batch(ctx, n_reads, n_samples);
    if (rc) return rc;
    if (n_samples && (rc = pxg_h2d_big(ctx, ctx->raw.p, raw_arena, (size_t)n_samples * sizeof(int16_t), ctx->stream)))
        return rc;
    if ((rc = pxg_h2d_meta(ctx, 1, 0, ctx->offsets.p, raw_offsets, (size_t)(n_reads + 1) * sizeof(int64_t), ctx->stream)) ||
        (rc = pxg_h2d_meta(ctx, 1, 1, ctx->calib.p, calib, (size_t)n_reads * sizeof(pxg_calib), ctx->stream)))
        return rc;
    ctx->have_inject = scale_shift_or_null != nullptr;
    if (ctx->have_inject &&
        (rc = pxg_h2d_meta(ctx, 1, 2, ctx->inject.p, scale_shift_or_null, (size_t)n_reads * 2 * sizeof(float), ctx->stream)))
        return rc;
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host buffers may be reused on return
    ctx->n_reads = n_reads;
    ctx->n_samples = n_samples;
    ctx->longest_read = longest_of(raw_offsets, n_reads);
    ctx->resident_limit = 0;
    rate_range(calib, n_reads, ctx->rate_min, ctx->rate_max);
    return PXG_OK;
}

extern "C" int pxg_batch_upload_tiled(pxg_ctx* ctx, int64_t n_reads, int64_t base_n, int64_t phase,
                                      const int16_t* base_arena, const int64_t* base_offsets,
                                      const pxg_calib* base_calib, const float* base_ss)
{
    if (!ctx) return PXG_E_INVALID;
    if (n_reads < 1 || base_n < 1 || phase < 0 || !base_offsets || !base_calib || !base_arena)
        return fail(ctx, PXG_E_INVALID, "pxg_batch_upload_tiled: bad arguments");
    for (int64_t i = 0; i < base_n; i++)
        if (base_offsets[i + 1] < base_offsets[i])
            return fail(ctx, PXG_E_INVALID, "raw_offsets must be non-decreasing");
    if (base_offsets[0] != 0) return fail(ctx, PXG_E_INVALID, "raw_offsets[0] must be 0");
    if (n_reads > (1LL << 30)) return fail(ctx, PXG_E_INVALID, "too many reads");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    ctx->n_reads = 0;
    ctx->polya_unsettled = false;     // results not downloaded by now are given up
    phase %= base_n;
    const int64_t base_samples = base_offsets[base_n];
    // per-read metadata of the tiled batch on the host (a few MB even for 1M reads)
    std::vector<int64_t> off((size_t)n_reads + 1);
    std::vector<pxg_calib> cal((size_t)n_reads);
    std::vector<float> ss(base_ss ? (size_t)n_reads * 2 : 0);
    off[0] = 0;
    for (int64_t j = 0, b = phase; j < n_reads; j++, b = (b + 1 == base_n ? 0 : b + 1)) {
        off[j + 1] = off[j] + (base_offsets[b + 1] - base_offsets[b]);
        cal[j] = base_calib[b];
        if (base_ss) { ss[2 * j] = base_ss[2 * b]; ss[2 * j + 1] = base_ss[2 * b + 1]; }
    }
    const int64_t n_samples = off[n_reads];
    int rc = reserve_batch(ctx, n_reads, n_samples);
    if (rc) return rc;
    DevBuf<int16_t> base;                                   // the distinct reads, once over PCIe
    if ((rc = pxg_reserve(ctx, base, (size_t)base_samples + 64))) return rc;
    hipStream_t st = ctx->stream;
    hipError_t e = pxg_h2d_big(ctx, base.p, base_arena, (size_t)base_samples * sizeof(int16_t), st) == PXG_OK ? hipSuccess : hipErrorUnknown;
    // cyclic replication: [phase .. base_n) then whole copies of the base set
    int64_t at = 0, src = base_offsets[phase];
    while (e == hipSuccess && at < n_samples) {
        const int64_t len = std::min(base_samples - src, n_samples - at);
        if (len > 0)
            e = hipMemcpyAsync(ctx->raw.p + at, base.p + src, (size_t)len * sizeof(int16_t),
                               hipMemcpyDeviceToDevice, st);
        at += len;
        src = 0;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->offsets.p, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->calib.p, cal.data(), cal.size() * sizeof(pxg_calib), hipMemcpyHostToDevice, st);
    ctx->have_inject = base_ss != nullptr;
    if (e == hipSuccess && base_ss)
        e = hipMemcpyAsync(ctx->inject.p, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(base.p);
    if (e != hipSuccess) return fail(ctx, PXG_E_HIP, std::string("pxg_batch_upload_tiled: ") + hipGetErrorString(e));
    ctx->n_reads = n_reads;
    ctx->n_samples = n_samples;
    ctx->longest_read = longest_of(base_offsets, base_n);
    ctx->resident_limit = 0;
    rate_range(base_calib, base_n, ctx->rate_min, ctx->rate_max);
    return PXG_OK;
}

static int check_batch_args(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                            const int64_t* raw_offsets, const pxg_calib* calib, const char* who)
{
    if (n_reads < 0 || (n_reads > 0 && (!raw_offsets || !calib)))
        return fail(ctx, PXG_E_INVALID, std::string(who) + ": bad arguments");
    for (int64_t i = 0; i < n_reads; i++)
        if (raw_offsets[i + 1] < raw_offsets[i])
            return fail(ctx, PXG_E_INVALID, "raw_offsets must be non-decreasing");
    if (n_reads > 0 && raw_offsets[0] != 0) return fail(ctx, PXG_E_INVALID, "raw_offsets[0] must be 0");
    if (n_reads > 0 && raw_offsets[n_reads] > 0 && !raw_arena)
        return fail(ctx, PXG_E_INVALID, "raw_arena is null");
    if (n_reads > (1LL << 30)) return fail(ctx, PXG_E_INVALID, "too many reads");
    return PXG_OK;
}

extern "C" int pxg_batch_stage(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                               const int64_t* raw_offsets, const pxg_calib* calib,
                               const float* scale_shift_or_null)
{
    return pxg_batch_stage_prefix(ctx, n_reads, raw_arena, raw_offsets, calib, scale_shift_or_null, 0);
}

// The first `limit` samples of every read (0: all of them), the arena laid out as if all had come: reads not
// longer than the limit travel in one copy with their neighbours, a longer one ends the run.
static int copy_read_prefixes(pxg_ctx* ctx, hipStream_t cs, int16_t* dst, const int16_t* src, const int64_t* off,
                              int64_t n, int64_t limit)
{
    int64_t run0 = 0;
    auto flush = [&](int64_t end) -> int {
        if (end > run0)
            return pxg_h2d_big(ctx, dst + run0, src + run0, (size_t)(end - run0) * sizeof(int16_t), cs);
        return PXG_OK;
    };
    int rc;
    if (limit > 0)
        for (int64_t r = 0; r < n; r++)
            if (off[r + 1] - off[r] > limit) {
                if ((rc = flush(off[r] + limit))) return rc;
                run0 = off[r + 1];
            }
    return flush(off[n]);
}

// the stages a resident batch staged with a prefix limit cannot serve (they read whole reads)
static int need_whole_reads(pxg_ctx* ctx, const char* who)
{
    if (ctx->resident_limit > 0 && ctx->longest_read > ctx->resident_limit) {
        pxg_set_err(ctx, std::string(who) + ": the resident batch was staged with a prefix limit (pxg_batch_stage_prefix) "
                                            "shorter than its longest read");
        return PXG_E_STATE;
    }
    return PXG_OK;
}

extern "C" int pxg_batch_stage_prefix(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                                      const int64_t* raw_offsets, const pxg_calib* calib,
                                      const float* scale_shift_or_null, int64_t prefix_limit)
{
    if (!ctx) return PXG_E_INVALID;
    if (prefix_limit < 0) return fail(ctx, PXG_E_INVALID, "pxg_batch_stage_prefix: negative limit");
    int rc = check_batch_args(ctx, n_reads, raw_arena, raw_offsets, calib, "pxg_batch_stage");
    if (rc) return rc;
    if (n_reads == 0) return fail(ctx, PXG_E_INVALID, "pxg_batch_stage: empty batch");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    auto& sp = ctx->spare;
    sp.staged = false;
    const int64_t n_samples = raw_offsets[n_reads];
    if ((rc = pxg_reserve(ctx, sp.raw, (size_t)n_samples + 64)) ||
        (rc = pxg_reserve(ctx, sp.offsets, (size_t)n_reads + 1)) ||
        (rc = pxg_reserve(ctx, sp.calib, (size_t)n_reads)) ||
        (rc = pxg_reserve(ctx, sp.inject, (size_t)n_reads * 2)))
        return rc;
    // the spare buffers were the resident ones before the last swap: the copy may start once the
    // last run that read them is done (not the run that is in flight on the resident ones)
    if (ctx->run_recorded[ctx->cur ^ 1])
        PXG_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_run_done[ctx->cur ^ 1], 0));
    hipStream_t cs = ctx->copy_stream;
    if (n_samples && (rc = copy_read_prefixes(ctx, cs, sp.raw.p, raw_arena, raw_offsets, n_reads, prefix_limit))) return rc;
    sp.limit = prefix_limit;
    if ((rc = pxg_h2d_meta(ctx, 0, 0, sp.offsets.p, raw_offsets, (size_t)(n_reads + 1) * sizeof(int64_t), cs)) ||
        (rc = pxg_h2d_meta(ctx, 0, 1, sp.calib.p, calib, (size_t)n_reads * sizeof(pxg_calib), cs)))
        return rc;
    sp.have_inject = scale_shift_or_null != nullptr;
    if (sp.have_inject &&
        (rc = pxg_h2d_meta(ctx, 0, 2, sp.inject.p, scale_shift_or_null, (size_t)n_reads * 2 * sizeof(float), cs)))
        return rc;
    PXG_HIP(ctx, hipEventRecord(ctx->ev_staged, cs));
    sp.n_reads = n_reads;
    sp.n_samples = n_samples;
    ctx->spare_longest_read = longest_of(raw_offsets, n_reads);
    rate_range(calib, n_reads, sp.rate_min, sp.rate_max);
    sp.staged = true;
    return PXG_OK;
}

// pxg_batch_stage with the samples as encoded bytes (include/pxg.h): the bytes and the chunk
// records cross the link, a decode kernel on the copy stream fills the spare slot's int16 arena.
extern "C" int pxg_batch_stage_z(pxg_ctx* ctx, int64_t n_reads, const uint8_t* z, int64_t z_bytes,
                                 const pxg_z_chunk* chunks, int64_t n_chunks, int64_t data_base,
                                 int64_t dst_base, const int64_t* raw_offsets, const pxg_calib* calib,
                                 const float* scale_shift_or_null)
{
    return pxg_batch_stage_z_prefix(ctx, n_reads, z, z_bytes, chunks, n_chunks, data_base, dst_base, raw_offsets, calib,
                                    scale_shift_or_null, 0);
}

// Byte ranges of the chunks that hold the first `limit` samples of every read: a chunk is kept while its first
// sample lies less than `limit` samples into its read -- the rule k_z_decode applies (a foreign record may
// carry short chunks in the middle of a read; counting chunks instead of samples would then leave chunks the
// kernel decodes uncopied).  false: some chunk spans two reads (the encoder never writes one, the records may
// still say so) -- the caller then copies everything.
static bool z_prefix_ranges(const pxg_z_chunk* chunks, int64_t n_chunks, int64_t data_base, int64_t z_bytes,
                            const int64_t* off, int64_t n, int64_t limit, std::vector<std::pair<int64_t, int64_t>>& ranges)
{
    auto byte_of = [&](int64_t g) { return g < n_chunks ? chunks[g].data_off - data_base : z_bytes; };
    int64_t g = 0, run0 = 0;
    for (int64_t r = 0; r < n; r++) {
        const int64_t len = off[r + 1] - off[r];
        int64_t at = 0, first_dropped = -1;                 // samples of the read before chunk g
        while (at < len && g < n_chunks) {
            if (at >= limit && first_dropped < 0) first_dropped = g;
            at += chunks[g++].len;
        }
        if (at != len) return false;
        if (first_dropped >= 0) {
            if (byte_of(first_dropped) > run0) ranges.emplace_back(run0, byte_of(first_dropped));
            run0 = byte_of(g);
        }
    }
    if (z_bytes > run0) ranges.emplace_back(run0, z_bytes);
    return g == n_chunks;
}

extern "C" int pxg_batch_stage_z_prefix(pxg_ctx* ctx, int64_t n_reads, const uint8_t* z, int64_t z_bytes,
                                        const pxg_z_chunk* chunks, int64_t n_chunks, int64_t data_base,
                                        int64_t dst_base, const int64_t* raw_offsets, const pxg_calib* calib,
                                        const float* scale_shift_or_null, int64_t prefix_limit)
{
    if (!ctx) return PXG_E_INVALID;
    if (prefix_limit < 0) return fail(ctx, PXG_E_INVALID, "pxg_batch_stage_z_prefix: negative limit");
    int rc = check_batch_args(ctx, n_reads, (const int16_t*)z, raw_offsets, calib, "pxg_batch_stage_z");
    if (rc) return rc;
    if (n_reads == 0) return fail(ctx, PXG_E_INVALID, "pxg_batch_stage_z: empty batch");
    if (z_bytes < 0 || n_chunks < 0 || (n_chunks && !chunks)) return fail(ctx, PXG_E_INVALID, "pxg_batch_stage_z: bad arguments");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    auto& sp = ctx->spare;
    sp.staged = false;
    const int64_t n_samples = raw_offsets[n_reads];
    // the records come from a file: they must tile the arena and stay inside the byte stream
    // before a kernel is allowed to follow them
    if (pxg_z_check(n_chunks, chunks, data_base, z_bytes, dst_base, n_samples) != PXG_OK)
        return fail(ctx, PXG_E_INVALID, "pxg_batch_stage_z: the chunk records do not describe this byte "
                                        "stream / sample arena (truncated or corrupt bundle)");
    if ((rc = pxg_reserve(ctx, sp.raw, (size_t)n_samples + 64)) ||
        (rc = pxg_reserve(ctx, sp.offsets, (size_t)n_reads + 1)) ||
        (rc = pxg_reserve(ctx, sp.calib, (size_t)n_reads)) ||
        (rc = pxg_reserve(ctx, sp.inject, (size_t)n_reads * 2)) ||
        (rc = pxg_reserve(ctx, sp.z, (size_t)z_bytes + 16)) ||
        (rc = pxg_reserve(ctx, sp.zchunks, (size_t)n_chunks + 1)))
        return rc;
    if (ctx->run_recorded[ctx->cur ^ 1])
        PXG_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_run_done[ctx->cur ^ 1], 0));
    hipStream_t cs = ctx->copy_stream;
    std::vector<std::pair<int64_t, int64_t>> ranges;
    if (prefix_limit > 0 && !z_prefix_ranges(chunks, n_chunks, data_base, z_bytes, raw_offsets, n_reads, prefix_limit, ranges))
        prefix_limit = 0;
    if (prefix_limit == 0) {
        ranges.clear();
        if (z_bytes) ranges.emplace_back(0, z_bytes);
    }
    for (const auto& rg : ranges)
        if ((rc = pxg_h2d_big(ctx, sp.z.p + rg.first, z + rg.first, (size_t)(rg.second - rg.first), cs))) return rc;
    sp.limit = prefix_limit;
    if (n_chunks && (rc = pxg_h2d_big(ctx, sp.zchunks.p, chunks, (size_t)n_chunks * sizeof(pxg_z_chunk), cs))) return rc;
    if ((rc = pxg_h2d_meta(ctx, 0, 0, sp.offsets.p, raw_offsets, (size_t)(n_reads + 1) * sizeof(int64_t), cs)) ||
        (rc = pxg_h2d_meta(ctx, 0, 1, sp.calib.p, calib, (size_t)n_reads * sizeof(pxg_calib), cs)))
        return rc;
    sp.have_inject = scale_shift_or_null != nullptr;
    if (sp.have_inject &&
        (rc = pxg_h2d_meta(ctx, 0, 2, sp.inject.p, scale_shift_or_null, (size_t)n_reads * 2 * sizeof(float), cs)))
        return rc;
    if ((rc = pxg_launch_z_decode(ctx, cs, n_chunks, sp.z.p, z_bytes, sp.zchunks.p, data_base, dst_base, sp.raw.p,
                                  sp.offsets.p, n_reads, prefix_limit)))
        return rc;
    PXG_HIP(ctx, hipGetLastError());
    PXG_HIP(ctx, hipEventRecord(ctx->ev_staged, cs));
    sp.n_reads = n_reads;
    sp.n_samples = n_samples;
    ctx->spare_longest_read = longest_of(raw_offsets, n_reads);
    rate_range(calib, n_reads, sp.rate_min, sp.rate_max);
    sp.staged = true;
    return PXG_OK;
}

extern "C" int pxg_batch_swap(pxg_ctx* ctx)
{
    if (!ctx) return PXG_E_INVALID;
    auto& sp = ctx->spare;
    if (!sp.staged) return fail(ctx, PXG_E_STATE, "pxg_batch_swap: nothing staged");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    PXG_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));      // host arrays are free again
    PXG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_staged, 0));
    std::swap(ctx->raw, sp.raw);
    std::swap(ctx->offsets, sp.offsets);
    std::swap(ctx->calib, sp.calib);
    std::swap(ctx->inject, sp.inject);
    ctx->cur ^= 1;
    ctx->have_inject = sp.have_inject;
    ctx->n_reads = 0;
    ctx->polya_unsettled = false;     // results not downloaded by now are given up
    int rc = reserve_batch(ctx, sp.n_reads, sp.n_samples);     // per-batch intermediates
    if (rc) return rc;
    ctx->n_reads = sp.n_reads;
    ctx->n_samples = sp.n_samples;
    ctx->longest_read = ctx->spare_longest_read;
    ctx->resident_limit = sp.limit;
    ctx->rate_min = sp.rate_min;
    ctx->rate_max = sp.rate_max;
    sp.staged = false;
    return PXG_OK;
}

extern "C" int pxg_host_register(pxg_ctx* ctx, void* ptr, size_t bytes)
{
    if (!ctx || !ptr || !bytes) return PXG_E_INVALID;
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    PXG_HIP(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return PXG_OK;
}

extern "C" int pxg_host_unregister(pxg_ctx* ctx, void* ptr)
{
    if (!ptr) return PXG_E_INVALID;
    // (page locks are process-global: a range can be released after the context that registered it is gone --
    //  a session that outlives its context must not leave a registration behind on memory it is about to unmap)
    if (!ctx) {
        if (hipHostUnregister(ptr) == hipSuccess) return PXG_OK;
        (void)hipGetLastError();
        return PXG_E_HIP;
    }
    PXG_HIP(ctx, hipHostUnregister(ptr));
    return PXG_OK;
}

#define PXG_POLYA_BESIDE_MAX 20000

extern "C" int pxg_batch_run(pxg_ctx* ctx, uint32_t stage_mask)
{
    if (!ctx) return PXG_E_INVALID;
    const int64_t n = ctx->n_reads;
    if (n <= 0) return fail(ctx, PXG_E_STATE, "pxg_batch_run: no resident batch");
    if ((stage_mask & PXG_STAGE_POLYA) && !(stage_mask & PXG_STAGE_SEGMENT))
        return fail(ctx, PXG_E_INVALID, "poly(A) stage needs the segment stage");
    if ((stage_mask & PXG_STAGE_BARCODE) && !(stage_mask & PXG_STAGE_SEGMENT))
        return fail(ctx, PXG_E_INVALID, "barcode stage needs the segment stage");
    if ((stage_mask & PXG_STAGE_SEGMENT) && !(stage_mask & PXG_STAGE_SCALER) && !ctx->have_inject)
        return fail(ctx, PXG_E_INVALID, "segment stage needs the scaler stage or injected scaling");
    if ((stage_mask & PXG_STAGE_POLYA) && need_whole_reads(ctx, "pxg_batch_run (poly(A) stage)")) return PXG_E_STATE;
    if (ctx->resident_limit > 0 && ctx->resident_limit < std::max(ctx->cfg.scaler_length, ctx->cfg.segmentation_scan_limit) &&
        need_whole_reads(ctx, "pxg_batch_run (the scaler reads scaler_length, the segmentation segmentation_scan_limit samples)"))
        return PXG_E_STATE;
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    memset(ctx->ev_used, 0, sizeof(ctx->ev_used));
    memset(ctx->launches, 0, sizeof(ctx->launches));
    int rc;
    pxg_timer_begin(ctx, PXG_T_TOTAL);
    if ((rc = pxg_launch_reset_batch(ctx, n))) return rc;
    const int32_t* order = nullptr;
    if (ctx->length_order && (stage_mask & (PXG_STAGE_SCALER | PXG_STAGE_SEGMENT))) {
        if ((rc = pxg_reserve(ctx, ctx->order, (size_t)n)) ||
            (rc = pxg_launch_length_order(ctx, n, ctx->offsets.p, ctx->order.p))) return rc;
        order = ctx->order.p;
    }

    if (stage_mask & PXG_STAGE_SCALER) {
        pxg_timer_begin(ctx, PXG_T_HEAD_POOL);
        if ((rc = pxg_launch_head_pool(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p, ctx->head.p,
                                       ctx->status.p))) return rc;
        pxg_timer_end(ctx, PXG_T_HEAD_POOL);
        if (ctx->have_inject) {
            PXG_HIP(ctx, hipMemcpyAsync(ctx->ss.p, ctx->inject.p, (size_t)n * 2 * sizeof(float),
                                        hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            if ((rc = pxg_launch_compact_scaler(ctx, n, ctx->status.p, order, ctx->idx_scaler.p, ctx->counters.p)))
                return rc;
            pxg_timer_begin(ctx, PXG_T_SCALER_LSTM);
            if ((rc = pxg_launch_scaler_lstm(ctx, n, ctx->idx_scaler.p, ctx->counters.p, ctx->head.p,
                                             ctx->pred.p, ctx->offsets.p))) return rc;
            pxg_timer_end(ctx, PXG_T_SCALER_LSTM);
            if ((rc = pxg_launch_scaler_transform(ctx, n, ctx->pred.p, ctx->ss.p, ctx->status.p,
                                                  ctx->idx_scaler.p, ctx->counters.p))) return rc;
        }
    } else if (ctx->have_inject) {
        PXG_HIP(ctx, hipMemcpyAsync(ctx->ss.p, ctx->inject.p, (size_t)n * 2 * sizeof(float),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (stage_mask & PXG_STAGE_SEGMENT) {
        pxg_timer_begin(ctx, PXG_T_SEGMENT);
        // with the scaler stage in the same run K1's block means of the head are still in HBM
        const float* head = (stage_mask & PXG_STAGE_SCALER) ? ctx->head.p : nullptr;
        if ((rc = pxg_launch_segment_raw(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p, ctx->ss.p,
                                         head, ctx->status.p, ctx->segs.p, order))) return rc;
        pxg_timer_end(ctx, PXG_T_SEGMENT);
        if (stage_mask & PXG_STAGE_POLYA)     // (the block means of a scan call, and K6 itself, may be taken from here on)
            PXG_HIP(ctx, hipEventRecord(ctx->ev_segmented, ctx->stream));
    }
    ctx->polya_ran = false;
    ctx->scan_gate_set = false;
    // K6 beside the barcode kernels: everything it reads (samples, scaling, status, segments) is final behind K3
    // (measured, profiles/r06/ab_k6_beside_k5*.txt: 10 000 reads -- K6's 2 500 waves are one launch round, the kernel
    //  lasts as long as its longest windows and that tail now hides under K5a / K5b: `full` 14.8 -> 14.5-14.6 ms; at
    //  100 000 reads K6 is throughput-bound and the issue-bound LSTM kernels lose more than K6 gains: 120 -> 123 ms --
    //  so only batches of up to PXG_POLYA_BESIDE_MAX reads take the second stream)
    const bool polya_beside = (stage_mask & PXG_STAGE_POLYA) && (stage_mask & PXG_STAGE_BARCODE) && ctx->polya_overlap &&
                              ctx->polya_stream != nullptr && n <= PXG_POLYA_BESIDE_MAX;
    auto run_polya = [&]() -> int {
        int prc;
        if ((prc = pxg_reserve(ctx, ctx->polya_out, (size_t)n * 8))) return prc;
        pxg_timer_begin(ctx, PXG_T_POLYA);
        if ((prc = pxg_launch_polya(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p, ctx->ss.p,
                                    ctx->status.p, ctx->segs.p, ctx->polya_out.p, ctx->spikes))) return prc;
        pxg_timer_end(ctx, PXG_T_POLYA);
        ctx->polya_ran = true;
        return PXG_OK;
    };
    if (polya_beside) {
        hipStream_t main = ctx->stream;
        if (hipStreamWaitEvent(ctx->polya_stream, ctx->ev_segmented, 0) != hipSuccess) return fail(ctx, PXG_E_HIP, "poly(A) stream: wait");
        ctx->stream = ctx->polya_stream;       // (every launch helper and timer takes ctx->stream; the caller holds the run lock)
        rc = run_polya();
        const hipError_t e = rc == PXG_OK ? hipEventRecord(ctx->ev_polya_done, ctx->polya_stream) : hipSuccess;
        if (rc != PXG_OK) (void)hipStreamSynchronize(ctx->polya_stream);
        ctx->stream = main;
        if (rc) return rc;
        if (e != hipSuccess) return fail(ctx, PXG_E_HIP, "poly(A) stream: event");
    }
    if (stage_mask & PXG_STAGE_BARCODE) {
        pxg_timer_begin(ctx, PXG_T_BARCODE_WINDOW);
        if ((rc = pxg_launch_barcode_window_raw(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p,
                                                ctx->ss.p, ctx->status.p, ctx->segs.p, ctx->win.p,
                                                ctx->idx_demux.p, ctx->counters.p + 1))) return rc;
        pxg_timer_end(ctx, PXG_T_BARCODE_WINDOW);
        if ((rc = pxg_launch_demux_lstm(ctx, n, ctx->idx_demux.p, ctx->counters.p + 1, ctx->win.p,
                                        ctx->bidir.p, ctx->probs.p, PXG_T_DEMUX_BIDIR,
                                        PXG_T_DEMUX_TOP))) return rc;
    }
    if (stage_mask & PXG_STAGE_POLYA) {
        if (ctx->scan_overlap) {        // a window scan called next may start here: behind the barcode kernels, beside K6 (or its tail)
            PXG_HIP(ctx, hipEventRecord(ctx->ev_scan_gate, ctx->stream));
            ctx->scan_gate_set = true;
        }
        if (!polya_beside) {
            if ((rc = run_polya())) return rc;
        } else {
            PXG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_polya_done, 0));      // the records need K6's output
        }
    }
    ctx->polya_unsettled = ctx->polya_ran;
    ctx->last_stage_mask = stage_mask;
    pxg_timer_begin(ctx, PXG_T_FINALIZE);
    if ((rc = pxg_launch_finalize(ctx, n, stage_mask))) return rc;
    pxg_timer_end(ctx, PXG_T_FINALIZE);
    pxg_timer_end(ctx, PXG_T_TOTAL);
    PXG_HIP(ctx, hipEventRecord(ctx->ev_run_done[ctx->cur], ctx->stream));
    ctx->run_recorded[ctx->cur] = true;
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}

// the time-sliced LSTM kernels raise a device flag instead of spinning forever when a
// hand-over never arrives (cannot happen with all workgroups resident; belt and braces)
static int check_timeslice_flag(pxg_ctx* ctx)
{
    if (!ctx->lstm_err.p || !ctx->timeslice_used) return PXG_OK;
    int flag = 0;
    PXG_HIP(ctx, hipMemcpyAsync(&flag, ctx->lstm_err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timeslice_used = false;
    if (flag) {
        (void)hipMemsetAsync(ctx->lstm_err.p, 0, sizeof(int), ctx->stream);
        return fail(ctx, PXG_E_HIP, "time-sliced LSTM kernel: tile hand-over timed out");
    }
    return PXG_OK;
}

// K6's first pass lists the reads whose inspection window needed more event rows than it
// hands out; they are re-run with what they asked for and the records rebuilt, before any
// result leaves the device.  No read listed (every batch but the odd one): one 8-byte copy.
static int settle_polya(pxg_ctx* ctx)
{
    if (!ctx->polya_unsettled) return PXG_OK;
    ctx->polya_unsettled = false;
    int64_t retried = 0;
    int rc = pxg_polya_settle(ctx, ctx->n_reads, ctx->longest_read, ctx->raw.p, ctx->offsets.p, ctx->calib.p,
                              ctx->ss.p, ctx->status.p, ctx->segs.p, ctx->polya_out.p, ctx->spikes, &retried,
                              &ctx->spike_rows);
    if (rc || !retried) return rc;
    return pxg_launch_finalize(ctx, ctx->n_reads, ctx->last_stage_mask);
}

extern "C" int pxg_batch_sync(pxg_ctx* ctx)
{
    if (!ctx) return PXG_E_INVALID;
    int rc = settle_polya(ctx);
    if (rc) return rc;
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_timeslice_flag(ctx);
}

int pxg_d2h_sync(pxg_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return PXG_OK;
    bool direct = bytes < (256u << 10);                 // small copies are staged by the runtime itself
    if (!direct) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess) direct = at.type == hipMemoryTypeHost;    // page-locked by the caller
        else (void)hipGetLastError();                   // (an unknown pointer is an error state of the runtime: clear it)
    }
    if (direct) {
        PXG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PXG_OK;
    }
    std::lock_guard<std::mutex> bounce_lock(ctx->mt_bounce);
    if (ctx->h_bounce_bytes < bytes) {
        if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
        ctx->h_bounce = nullptr;
        ctx->h_bounce_bytes = 0;
        const size_t want = bytes + bytes / 4;
        if (hipHostMalloc(&ctx->h_bounce, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, PXG_E_NOMEM, "page-locked bounce buffer");
        }
        ctx->h_bounce_bytes = want;
    }
    PXG_HIP(ctx, hipMemcpyAsync(ctx->h_bounce, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->h_bounce, bytes);
    return PXG_OK;
}

// Host -> device of a big array on stream `st`.  Completion semantics: a PAGEABLE source has been consumed when the call
// returns (like a pageable hipMemcpyAsync); a page-locked source (hipHostRegister by the caller, hipHostMalloc) is ONE
// asynchronous DMA and must stay untouched until `st` has passed it -- every caller of this function waits for its
// stream (or records an event on it) before the source is reused.  A PAGEABLE source of
// 1 MB or more is NOT handed to the runtime: ROCm 7.2 page-locks the caller's range in place for such a copy
// ("Locking to pool ... hostMem = <user address>", hsa_amd_memory_lock) and lets the SDMA engine read the user's
// pages, and when that range lies in the brk heap and overlaps pages that were hipHostRegister'ed and
// unregistered earlier in the process, the engine faults inside the just-locked range -- "Memory access fault by
// GPU ... on address <heap address>. Reason: Unknown.", the whole process gone (caught with the runtime's API /
// copy log in round 5: profiles/r05/fault_hunt.md; round 4's one unexplained fault has this signature).  So the
// bytes travel through two page-locked chunks of the context: the GPU never addresses pageable user memory.
#ifndef PXG_UP_CHUNK
#define PXG_UP_CHUNK (8u << 20)
#endif
#define PXG_UP_THREADS 2
#define PXG_UP_MAX_THREADS 8
int pxg_h2d_big(pxg_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (!bytes) return PXG_OK;
    // (PXG_H2D_RUNTIME_LOCKS=1: the round-4 behaviour, for tools/heap_pin_fault.py to show what it leads to)
    if (bytes < (512u << 10) || getenv("PXG_H2D_RUNTIME_LOCKS")) {       // staged by the runtime through ITS page-locked buffer
        PXG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return PXG_OK;
    }
    // Page-locked means the WHOLE range [src, src + bytes) lies in one registration: a range that starts inside a
    // registered block and runs past its end would reach the runtime as one DMA over pageable pages -- the
    // in-place-lock path this function exists to avoid.  Both ends are asked; a device or managed source is the
    // runtime's own business (plain asynchronous copy).
    hipPointerAttribute_t at, at_end;
    if (hipPointerGetAttributes(&at, src) == hipSuccess) {
        if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) {
            PXG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
            return PXG_OK;
        }
        if (at.type == hipMemoryTypeHost) {             // page-locked by the caller -- at its first byte
            bool whole = false;
            if (hipPointerGetAttributes(&at_end, (const char*)src + bytes - 1) == hipSuccess)
                whole = at_end.type == hipMemoryTypeHost;
            else
                (void)hipGetLastError();
            if (whole) {
                PXG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
                return PXG_OK;
            }
        }
    } else {
        (void)hipGetLastError();                        // (an unknown pointer is an error state of the runtime: clear it)
    }
    std::lock_guard<std::mutex> up_lock(ctx->mt_up);
    // a big array (a whole batch's samples) is cut into PXG_UP_THREADS stretches, each filled by a host thread of
    // its own through its own pair of chunks: one thread's memcpy runs at ~13 GB/s, the link takes 57
    static const int want_threads = [] { const char* e = getenv("PXG_UP_THREADS"); const int v = e ? atoi(e) : PXG_UP_THREADS; return v < 1 ? 1 : (v > PXG_UP_MAX_THREADS ? PXG_UP_MAX_THREADS : v); }();
    const int nt = bytes >= (64u << 20) ? want_threads : 1;
    for (int k = 0; k < 2 * nt; k++) {
        if (ctx->h_up[k]) continue;
        if (hipHostMalloc(&ctx->h_up[k], PXG_UP_CHUNK, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, PXG_E_NOMEM, "page-locked upload chunks");
        }
        ctx->up_busy[k] = false;
    }
    std::atomic<int> failed{ 0 };
    auto stretch = [&](int t) {
        (void)hipSetDevice(ctx->device);
        const size_t part = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
        const size_t lo = std::min(bytes, part * (size_t)t), hi = std::min(bytes, lo + part);
        size_t done = lo;
        int k = 2 * t;
        while (done < hi && !failed.load()) {
            const size_t len = std::min<size_t>(PXG_UP_CHUNK, hi - done);
            if (ctx->up_busy[k] && hipEventSynchronize(ctx->ev_up[k]) != hipSuccess) { failed.store(1); break; }
            memcpy(ctx->h_up[k], (const char*)src + done, len);
            if (hipMemcpyAsync((char*)dst + done, ctx->h_up[k], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(ctx->ev_up[k], st) != hipSuccess) { failed.store(1); break; }
            ctx->up_busy[k] = true;
            done += len;
            k = 2 * t + ((k + 1) & 1);
        }
    };
    if (nt == 1) {
        stretch(0);
    } else {
        std::vector<std::thread> workers;
        try {
            for (int t = 1; t < nt; t++) workers.emplace_back(stretch, t);
        } catch (const std::exception&) {
            failed.store(2);                           // (no thread to be had: the stretches nobody took are not copied)
        }
        stretch(0);
        for (auto& w : workers) w.join();
    }
    if (failed.load()) {
        (void)hipGetLastError();
        return fail(ctx, PXG_E_HIP, "chunked host -> device copy");
    }
    return PXG_OK;
}

int pxg_h2d_meta(pxg_ctx* ctx, int set, int which, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (!bytes) return PXG_OK;
    auto& m = ctx->h_meta[set][which];
    if (m.cap < bytes) {
        if (m.p) (void)hipHostFree(m.p);
        m.p = nullptr;
        m.cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&m.p, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, PXG_E_NOMEM, "page-locked mirror of a batch's small arrays");
        }
        m.cap = want;
    }
    memcpy(m.p, src, bytes);
    PXG_HIP(ctx, hipMemcpyAsync(dst, m.p, bytes, hipMemcpyHostToDevice, st));
    return PXG_OK;
}

extern "C" int pxg_batch_download(pxg_ctx* ctx, pxg_read_result* out)
{
    if (!ctx || (!out && ctx->n_reads)) return PXG_E_INVALID;
    if (ctx->n_reads <= 0) return PXG_OK;
    int rc = settle_polya(ctx);
    if (rc) return rc;
    if ((rc = pxg_d2h_sync(ctx, out, ctx->results.p, (size_t)ctx->n_reads * sizeof(pxg_read_result)))) return rc;
    return check_timeslice_flag(ctx);
}

extern "C" int pxg_batch_download_samples(pxg_ctx* ctx, int16_t* out)
{
    if (!ctx || (!out && ctx->n_samples)) return PXG_E_INVALID;
    if (ctx->n_reads <= 0) return fail(ctx, PXG_E_STATE, "pxg_batch_download_samples: no resident batch");
    if (need_whole_reads(ctx, "pxg_batch_download_samples")) return PXG_E_STATE;
    if (ctx->n_samples <= 0) return PXG_OK;
    return pxg_d2h_sync(ctx, out, ctx->raw.p, (size_t)ctx->n_samples * sizeof(int16_t));
}

extern "C" int pxg_batch_download_windows(pxg_ctx* ctx, float* out)
{
    if (!ctx || !out) return PXG_E_INVALID;
    if (ctx->n_reads <= 0) return fail(ctx, PXG_E_STATE, "pxg_batch_download_windows: no resident batch");
    if (!(ctx->last_stage_mask & PXG_STAGE_BARCODE))
        return fail(ctx, PXG_E_STATE, "pxg_batch_download_windows: the last run had no barcode stage");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    return pxg_d2h_sync(ctx, out, ctx->win.p, (size_t)ctx->n_reads * ctx->cfg.signal_trim_length * sizeof(float));
}

extern "C" int pxg_batch_download_spikes(pxg_ctx* ctx, int64_t cap_rows, pxg_polya_spike* out, int64_t* offsets)
{
    if (!ctx || (!offsets && ctx->n_reads)) return PXG_E_INVALID;
    if (ctx->n_reads <= 0) { if (offsets) offsets[0] = 0; return PXG_OK; }
    if (!ctx->polya_ran) return fail(ctx, PXG_E_STATE, "the last run had no poly(A) stage");
    PXG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_polya(ctx);
    if (rc) return rc;
    return pxg_polya_collect_spikes(ctx, ctx->n_reads, ctx->polya_out.p, ctx->spikes, ctx->spike_rows, cap_rows,
                                    out, offsets);
}

extern "C" int pxg_batch_times(pxg_ctx* ctx, pxg_stage_times* out)
{
    if (!ctx || !out) return PXG_E_INVALID;
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < PXG_N_TIMERS; t++) {
        out->ms[t] = 0.0f;
        out->n_launches[t] = ctx->launches[t];
        if (ctx->ev_used[t]) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ctx->ev_start[t], ctx->ev_stop[t]) == hipSuccess)
                out->ms[t] = ms;
        }
    }
    return PXG_OK;
}

extern "C" int pxg_process_batch_ex(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                                    const int64_t* raw_offsets, const pxg_calib* calib, uint32_t stage_mask,
                                    pxg_batch_extras* x, pxg_read_result* out);

extern "C" int pxg_process_batch(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                                 const int64_t* raw_offsets, const pxg_calib* calib,
                                 const float* scale_shift_or_null, uint32_t stage_mask,
                                 pxg_read_result* out)
{
    if (!scale_shift_or_null)
        return pxg_process_batch_ex(ctx, n_reads, raw_arena, raw_offsets, calib, stage_mask, nullptr, out);
    pxg_batch_extras x;
    memset(&x, 0, sizeof(x));
    x.struct_bytes = sizeof(x);
    x.scale_shift_or_null = scale_shift_or_null;
    return pxg_process_batch_ex(ctx, n_reads, raw_arena, raw_offsets, calib, stage_mask, &x, out);
}

extern "C" int pxg_batch_unsplit_scan(pxg_ctx* ctx, const int64_t* first_sample, const int64_t* n_blocks,
                                      int32_t block_stride, int64_t cap_intervals, int64_t* out_intervals,
                                      int32_t* out_count, int64_t* out_total);

// ---- small calls share a batch -----------------------------------------------------------------------
// A call of at most PXG_MERGE_MAX_READS reads with plain inputs (int16 samples, no injected scaling, no window
// scan, no spike rows) joins the group that is waiting for the spare input slot.  The first caller of a group leads:
// it waits for the slot -- which is exactly as long as the pipeline is full (one batch computing, one staged), so
// groups grow with the load and a lone call starts at once --, takes everything that has gathered, stages the
// reads of all calls back to back (each call's arena is copied from where it lies), runs the batch, and hands
// every call its slice of the records.  Results are those of separate calls: every record is a function of its
// read alone (tests/test_gpu_parity.py::test_small_calls_are_merged_into_one_batch).
#define PXG_MERGE_MAX_READS 4096
#define PXG_MERGE_MAX_GROUP_READS 65536

static int merged_process(pxg_ctx* ctx, pxg_ctx::MergeItem& mine, uint32_t stage_mask, bool& bypass)
{
    auto& mq = ctx->merge;
    bypass = false;
    std::unique_lock<std::mutex> lk(mq.m);
    if (!mq.pending.empty() && mq.mask != stage_mask) {       // another stage mask is gathering: go alone
        bypass = true;
        return PXG_OK;
    }
    // a group never grows beyond what ONE call could stage (PXG_MERGE_MAX_GROUP_READS reads): a call that would
    // overfill it goes alone, behind the group
    if (mq.pending_reads + mine.n > PXG_MERGE_MAX_GROUP_READS) {
        bypass = true;
        return PXG_OK;
    }
    try {
        mq.pending.push_back(&mine);
    } catch (const std::exception& e) {
        return fail(ctx, PXG_E_NOMEM, std::string("merged calls: ") + e.what());
    }
    mq.pending_reads += mine.n;
    mq.mask = stage_mask;
    if (mq.leader) {
        mq.cv.wait(lk, [&] { return mine.done; });
        return mine.rc;
    }
    mq.leader = true;
    lk.unlock();
    std::unique_lock<std::mutex> stage_lock(ctx->mt_stage);   // <- the group grows while this waits
    lk.lock();
    std::vector<pxg_ctx::MergeItem*> items;
    items.swap(mq.pending);
    mq.pending_reads = 0;
    mq.leader = false;
    mq.groups++;
    mq.calls += (int64_t)items.size();
    lk.unlock();

    int rc = PXG_OK;
    std::unique_lock<std::mutex> run_lock(ctx->mt_run, std::defer_lock);
    try {       // (whatever happens, the calls that wait on this one are released below)
    do {
        int64_t n = 0, n_samples = 0, zb = 0, zc = 0;
        for (auto* it : items) {
            n += it->n;
            n_samples += it->off[it->n];
            if (!it->arena) { zb += it->z_bytes; zc += it->n_chunks; }
        }
        mq.h_off.resize((size_t)n + 1);
        mq.h_cal.resize((size_t)n);
        int64_t r = 0, s0 = 0;
        for (auto* it : items) {
            for (int64_t i = 0; i < it->n; i++) mq.h_off[(size_t)(r + i)] = s0 + it->off[i];
            memcpy(mq.h_cal.data() + r, it->cal, (size_t)it->n * sizeof(pxg_calib));
            r += it->n;
            s0 += it->off[it->n];
        }
        mq.h_off[(size_t)n] = n_samples;
        if ((rc = hipSetDevice(ctx->device) == hipSuccess ? PXG_OK : fail(ctx, PXG_E_HIP, "hipSetDevice"))) break;
        auto& sp = ctx->spare;
        sp.staged = false;
        if ((rc = pxg_reserve(ctx, sp.raw, (size_t)n_samples + 64)) || (rc = pxg_reserve(ctx, sp.offsets, (size_t)n + 1)) ||
            (rc = pxg_reserve(ctx, sp.calib, (size_t)n)) || (rc = pxg_reserve(ctx, sp.inject, (size_t)n * 2)) ||
            (zc && ((rc = pxg_reserve(ctx, sp.z, (size_t)zb + 16)) || (rc = pxg_reserve(ctx, sp.zchunks, (size_t)zc + 1)))))
            break;
        hipStream_t cs = ctx->copy_stream;
        if (ctx->run_recorded[ctx->cur ^ 1] &&
            hipStreamWaitEvent(cs, ctx->ev_run_done[ctx->cur ^ 1], 0) != hipSuccess) { rc = fail(ctx, PXG_E_HIP, "hipStreamWaitEvent"); break; }
        const int64_t limit = ((stage_mask & PXG_STAGE_POLYA) || zc) ? 0 : (int64_t)std::max(ctx->cfg.scaler_length, ctx->cfg.segmentation_scan_limit);
        s0 = 0;
        int64_t z0 = 0, c0 = 0;
        for (auto* it : items) {
            if (it->arena) {
                if (it->off[it->n] && (rc = copy_read_prefixes(ctx, cs, sp.raw.p + s0, it->arena, it->off, it->n, limit))) break;
            } else if (it->n_chunks) {
                // an encoded call: its bytes and chunk records behind the others', decoded with ITS bases into its
                // stretch of the arena (whole reads: the prefix rule is for calls that come alone)
                if ((it->z_bytes && pxg_h2d_big(ctx, sp.z.p + z0, it->z, (size_t)it->z_bytes, cs) != PXG_OK) ||
                    pxg_h2d_big(ctx, sp.zchunks.p + c0, it->chunks, (size_t)it->n_chunks * sizeof(pxg_z_chunk), cs) != PXG_OK) {
                    rc = fail(ctx, PXG_E_HIP, "merged stage: copy of encoded samples");
                    break;
                }
                if ((rc = pxg_launch_z_decode(ctx, cs, it->n_chunks, sp.z.p + z0, it->z_bytes, sp.zchunks.p + c0, it->data_base,
                                              it->dst_base, sp.raw.p + s0)))
                    break;
                z0 += it->z_bytes;
                c0 += it->n_chunks;
            }
            s0 += it->off[it->n];
        }
        if (rc) break;
        if ((rc = pxg_h2d_meta(ctx, 0, 0, sp.offsets.p, mq.h_off.data(), (size_t)(n + 1) * sizeof(int64_t), cs)) ||
            (rc = pxg_h2d_meta(ctx, 0, 1, sp.calib.p, mq.h_cal.data(), (size_t)n * sizeof(pxg_calib), cs)))
            break;
        if (hipEventRecord(ctx->ev_staged, cs) != hipSuccess) { rc = fail(ctx, PXG_E_HIP, "merged stage: event"); break; }
        sp.limit = limit;
        sp.have_inject = false;
        sp.n_reads = n;
        sp.n_samples = n_samples;
        ctx->spare_longest_read = longest_of(mq.h_off.data(), n);
        rate_range(mq.h_cal.data(), n, sp.rate_min, sp.rate_max);
        sp.staged = true;
        run_lock.lock();                              // the previous batch has all its results
        rc = pxg_batch_swap(ctx);
        stage_lock.unlock();                          // the next group (or call) may start its copy
        if (rc) break;
        if ((rc = pxg_batch_run(ctx, stage_mask))) break;
        mq.h_out.resize((size_t)n);
        if ((rc = pxg_batch_download(ctx, mq.h_out.data()))) break;
        r = 0;
        for (auto* it : items) {
            memcpy(it->out, mq.h_out.data() + r, (size_t)it->n * sizeof(pxg_read_result));
            r += it->n;
        }
    } while (0);
    } catch (const std::exception& e) {
        rc = fail(ctx, PXG_E_NOMEM, std::string("merged calls: ") + e.what());
    }
    // a failed group may have copies of the callers' arenas queued: nobody is released (and frees its arena) before
    // they have run
    if (rc != PXG_OK) (void)hipStreamSynchronize(ctx->copy_stream);
    if (run_lock.owns_lock()) run_lock.unlock();
    if (stage_lock.owns_lock()) stage_lock.unlock();
    lk.lock();
    for (auto* it : items) {
        it->rc = rc;
        it->done = true;
    }
    lk.unlock();
    mq.cv.notify_all();
    return rc;
}

extern "C" int pxg_merge_stats(pxg_ctx* ctx, int64_t* groups, int64_t* calls)
{
    if (!ctx || !groups || !calls) return PXG_E_INVALID;
    std::lock_guard<std::mutex> g(ctx->merge.m);
    *groups = ctx->merge.groups;
    *calls = ctx->merge.calls;
    return PXG_OK;
}

// One call per worker batch, from any number of host threads (include/pxg.h): the spare input
// slot belongs to one call from its copy to its swap (mt_stage), the resident batch and every
// per-batch intermediate from the swap to the last download (mt_run); always taken in this order.
extern "C" int pxg_process_batch_ex(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                                    const int64_t* raw_offsets, const pxg_calib* calib, uint32_t stage_mask,
                                    pxg_batch_extras* x, pxg_read_result* out)
{
    if (!ctx) return PXG_E_INVALID;
    if (x && x->struct_bytes != sizeof(pxg_batch_extras))
        return fail(ctx, PXG_E_INVALID, "pxg_batch_extras.struct_bytes does not match this library");
    if (x) { x->unsplit_total = 0; x->spike_total = 0; }
    if (n_reads == 0) {
        if (x && x->spike_offsets) x->spike_offsets[0] = 0;
        return PXG_OK;
    }
    if (!out) return fail(ctx, PXG_E_INVALID, "pxg_process_batch: out is null");
    const float* inject = x ? x->scale_shift_or_null : nullptr;
    int rc;
    const bool encoded = x && x->z;
    if (ctx->merge_small_calls && n_reads <= PXG_MERGE_MAX_READS && (raw_arena || encoded) && !inject &&
        !(x && (x->unsplit_first_sample || (x->spike_offsets && (stage_mask & PXG_STAGE_POLYA))))) {
        if ((rc = check_batch_args(ctx, n_reads, encoded ? (const int16_t*)x->z : raw_arena, raw_offsets, calib, "pxg_process_batch")))
            return rc;
        if (encoded && (x->z_bytes < 0 || x->n_chunks < 0 || (x->n_chunks && !x->chunks) ||
                        pxg_z_check(x->n_chunks, x->chunks, x->data_base, x->z_bytes, x->dst_base, raw_offsets[n_reads]) != PXG_OK))
            return fail(ctx, PXG_E_INVALID, "pxg_batch_stage_z: the chunk records do not describe this byte "
                                            "stream / sample arena (truncated or corrupt bundle)");
        pxg_ctx::MergeItem item = { n_reads, encoded ? nullptr : raw_arena, raw_offsets, calib, out,
                                    encoded ? x->z : nullptr, encoded ? x->z_bytes : 0, encoded ? x->chunks : nullptr,
                                    encoded ? x->n_chunks : 0, encoded ? x->data_base : 0, encoded ? x->dst_base : 0, PXG_OK, false };
        bool bypass = false;
        rc = merged_process(ctx, item, stage_mask, bypass);
        if (!bypass) return rc;
    }
    // PXG_TRACE=1: one stderr line per call with the milliseconds spent waiting for the spare slot,
    // copying, waiting for the resident batch, computing and downloading
    static const bool trace = getenv("PXG_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0.0;
    std::unique_lock<std::mutex> stage_lock(ctx->mt_stage);
    const double t1 = trace ? now() : 0.0;
    // signal_analyzer.py:347-349: without poly(A) and the chimera scan no stage reads a sample behind the
    // segmentation's scan limit (the scaler's head and the barcode window lie inside it) -- they stay on the host
    const int64_t limit = ((stage_mask & PXG_STAGE_POLYA) || (x && x->unsplit_first_sample)) ? 0
                          : (int64_t)std::max(ctx->cfg.scaler_length, ctx->cfg.segmentation_scan_limit);
    if (x && x->z)
        rc = pxg_batch_stage_z_prefix(ctx, n_reads, x->z, x->z_bytes, x->chunks, x->n_chunks, x->data_base, x->dst_base,
                                      raw_offsets, calib, inject, limit);
    else
        rc = pxg_batch_stage_prefix(ctx, n_reads, raw_arena, raw_offsets, calib, inject, limit);
    if (rc) return rc;
    const double t2 = trace ? now() : 0.0;
    std::unique_lock<std::mutex> run_lock(ctx->mt_run);      // the previous call has all its results
    const double t3 = trace ? now() : 0.0;
    rc = pxg_batch_swap(ctx);
    stage_lock.unlock();                                      // the next call may start its copy
    if (rc) return rc;
    const double t4 = trace ? now() : 0.0;
    if ((rc = pxg_batch_run(ctx, stage_mask))) return rc;
    const double t5 = trace ? now() : 0.0;
    int verdict = PXG_OK;
    if (x && x->unsplit_first_sample) {
        if ((rc = pxg_batch_unsplit_scan(ctx, x->unsplit_first_sample, x->unsplit_n_blocks, x->unsplit_block_stride,
                                         x->unsplit_cap, x->unsplit_intervals, x->unsplit_count, &x->unsplit_total)))
            return rc;
        if (x->unsplit_total > x->unsplit_cap) verdict = PXG_E_NOMEM;
    }
    if ((rc = pxg_batch_download(ctx, out))) return rc;
    if (x && x->spike_offsets && (stage_mask & PXG_STAGE_POLYA)) {
        rc = pxg_batch_download_spikes(ctx, x->spike_cap, x->spikes, x->spike_offsets);
        x->spike_total = x->spike_offsets[n_reads];
        if (rc == PXG_E_NOMEM) verdict = PXG_E_NOMEM;
        else if (rc) return rc;
    }
    if (trace)
        fprintf(stderr, "[pxg] call at %.1f: wait-spare %.2f stage %.2f wait-resident %.2f swap %.2f launch %.2f "
                        "results %.2f ms\n", t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, now() - t5);
    return verdict;
}

// ---------------------------------------------------------------------------
// stage hooks: host arrays in, host arrays out, through the same kernels
// ---------------------------------------------------------------------------
struct Scratch {               // RAII device temporaries for the hooks
    pxg_ctx* ctx = nullptr;
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    T* alloc(size_t n)
    {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T*)p;
    }
    template <typename T>
    T* put(const T* src, size_t n, hipStream_t s)
    {
        T* d = alloc<T>(n);
        // (a big pageable array of the caller never reaches the runtime: pxg_h2d_big; a failed copy reads as a failed allocation)
        if (d && n && pxg_h2d_big(ctx, d, src, n * sizeof(T), s) != PXG_OK) return nullptr;
        return d;
    }
};

// an early `return rc` behind the first asynchronous copy / launch of a call leaves work queued that reads buffers the
// caller may free next: the stream is drained on every exit the success path has not dismissed
struct StreamSyncOnError {
    hipStream_t st;
    bool armed = true;
    explicit StreamSyncOnError(hipStream_t s) : st(s) {}
    void dismiss() { armed = false; }
    ~StreamSyncOnError() { if (armed) (void)hipStreamSynchronize(st); }
};

// The scan calls enqueue on the scan stream when the last run left a gate (see pxg_common.h): for their duration
// ctx->stream IS the scan stream (every launch helper and timer takes ctx->stream), put back on every exit; `join`
// makes the main stream wait for what was enqueued, so that the downloads behind it see the scan's results.
struct ScanStreamScope {
    pxg_ctx* ctx;
    hipStream_t main;
    bool on = false;
    explicit ScanStreamScope(pxg_ctx* c) : ctx(c), main(c->stream)
    {
        if (c->scan_gate_set && c->scan_stream &&
            hipStreamWaitEvent(c->scan_stream, c->ev_segmented, 0) == hipSuccess) {
            c->stream = c->scan_stream;
            on = true;
        }
    }
    // the second gate: what is enqueued from here on starts with K6 (the block means before it: with the barcode kernels)
    int gate_polya() { return (!on || hipStreamWaitEvent(ctx->scan_stream, ctx->ev_scan_gate, 0) == hipSuccess) ? PXG_OK : PXG_E_HIP; }
    int join()
    {
        if (!on) return PXG_OK;
        on = false;
        const hipError_t e1 = hipEventRecord(ctx->ev_scan_done, ctx->scan_stream);
        ctx->stream = main;
        const hipError_t e2 = hipStreamWaitEvent(main, ctx->ev_scan_done, 0);
        return (e1 == hipSuccess && e2 == hipSuccess) ? PXG_OK : PXG_E_HIP;
    }
    ~ScanStreamScope()
    {
        if (on) {                       // an error exit: nothing of the scan may still be running when the caller goes on
            (void)hipStreamSynchronize(ctx->scan_stream);
            ctx->stream = main;
        }
    }
};

#define HOOK_BEGIN                                           \
    if (!ctx) return PXG_E_INVALID;                          \
    PXG_HIP(ctx, hipSetDevice(ctx->device));                 \
    Scratch S;                                               \
    S.ctx = ctx;
#define HOOK_CHECK(p) if (!(p)) return fail(ctx, PXG_E_NOMEM, "hook scratch allocation failed")
#define HOOK_GET(dst, src, n)                                                                         \
    do {                                                                                              \
        const int rc_get__ = pxg_d2h_sync(ctx, dst, src, (size_t)(n) * sizeof(*(dst)));               \
        if (rc_get__) return rc_get__;                                                                \
    } while (0)
#define HOOK_END                                             \
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));         \
    PXG_HIP(ctx, hipGetLastError());                         \
    return PXG_OK;

extern "C" int pxg_raw_to_pa(pxg_ctx* ctx, int64_t n, const int16_t* raw, const pxg_calib* calib,
                             float* out)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    int16_t* d_raw = S.put(raw, (size_t)n, ctx->stream);
    float* d_out = S.alloc<float>((size_t)n);
    HOOK_CHECK(d_raw && d_out);
    int rc = pxg_launch_raw_to_pa(ctx, n, d_raw, calib, d_out);
    if (rc) return rc;
    HOOK_GET(out, d_out, n);
    HOOK_END
}

extern "C" int pxg_head_pool(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                             const pxg_calib* calib, float* out, int32_t* status)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    const size_t width = (size_t)(ctx->cfg.scaler_length / ctx->cfg.stride);
    int16_t* d_raw = S.put(raw, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    pxg_calib* d_cal = S.put(calib, (size_t)n, ctx->stream);
    float* d_out = S.alloc<float>((size_t)n * width);
    int32_t* d_st = S.alloc<int32_t>((size_t)n);
    HOOK_CHECK(d_raw && d_off && d_cal && d_out && d_st);
    int rc = pxg_launch_head_pool(ctx, n, d_raw, d_off, d_cal, d_out, d_st);
    if (rc) return rc;
    HOOK_GET(out, d_out, (size_t)n * width);
    HOOK_GET(status, d_st, n);
    HOOK_END
}

extern "C" int pxg_scaler_lstm(pxg_ctx* ctx, int64_t n, const float* head, float* pred)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    const size_t width = (size_t)(ctx->cfg.scaler_length / ctx->cfg.stride);
    float* d_head = S.put(head, (size_t)n * width, ctx->stream);
    float* d_pred = S.alloc<float>((size_t)n * 2);
    HOOK_CHECK(d_head && d_pred);
    int rc = pxg_launch_scaler_lstm(ctx, n, nullptr, nullptr, d_head, d_pred);
    if (rc) return rc;
    HOOK_GET(pred, d_pred, (size_t)n * 2);
    HOOK_END
}

extern "C" int pxg_scaler_transform(pxg_ctx* ctx, int64_t n, const float* pred, float* scale_shift,
                                    int32_t* status)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    float* d_pred = S.put(pred, (size_t)n * 2, ctx->stream);
    float* d_ss = S.alloc<float>((size_t)n * 2);
    int32_t* d_st = S.alloc<int32_t>((size_t)n);
    HOOK_CHECK(d_pred && d_ss && d_st);
    int rc = pxg_launch_scaler_transform(ctx, n, d_pred, d_ss, d_st, nullptr, nullptr);
    if (rc) return rc;
    HOOK_GET(scale_shift, d_ss, (size_t)n * 2);
    HOOK_GET(status, d_st, n);
    HOOK_END
}

extern "C" int pxg_pool_scale(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                              const pxg_calib* calib, const float* scale_shift,
                              const int64_t* pooled_offsets, float* out)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    int16_t* d_raw = S.put(raw, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    pxg_calib* d_cal = S.put(calib, (size_t)n, ctx->stream);
    float* d_ss = S.put(scale_shift, (size_t)n * 2, ctx->stream);
    int64_t* d_poff = S.put(pooled_offsets, (size_t)n + 1, ctx->stream);
    float* d_out = S.alloc<float>((size_t)pooled_offsets[n]);
    HOOK_CHECK(d_raw && d_off && d_cal && d_ss && d_poff && d_out);
    int rc = pxg_launch_pool_scale(ctx, n, d_raw, d_off, d_cal, d_ss, d_poff, d_out);
    if (rc) return rc;
    HOOK_GET(out, d_out, (size_t)pooled_offsets[n]);
    HOOK_END
}

extern "C" int pxg_batch_pooled_signal(pxg_ctx* ctx, const int64_t* first, const int64_t* out_offsets, float* out)
{
    HOOK_BEGIN
    const int64_t n = ctx->n_reads;
    if (n <= 0) return fail(ctx, PXG_E_STATE, "pxg_batch_pooled_signal: no resident batch");
    if (!(ctx->last_stage_mask & (PXG_STAGE_SCALER | PXG_STAGE_SEGMENT)))    // (scale / shift: fitted or injected)
        return fail(ctx, PXG_E_STATE, "pxg_batch_pooled_signal: nothing has been run on the resident batch");
    if (!first || !out_offsets) return fail(ctx, PXG_E_INVALID, "pxg_batch_pooled_signal: bad arguments");
    for (int64_t i = 0; i < n; i++)
        if (out_offsets[i + 1] < out_offsets[i] || out_offsets[0] != 0)
            return fail(ctx, PXG_E_INVALID, "pxg_batch_pooled_signal: out_offsets must start at 0 and not decrease");
    const size_t total = (size_t)out_offsets[n];
    if (!total) return PXG_OK;
    if (!out) return fail(ctx, PXG_E_INVALID, "pxg_batch_pooled_signal: out is null");
    // a batch staged with a prefix limit holds only the first `resident_limit` samples of each read: a stretch that
    // ends behind them would be pooled from whatever the arena held before
    if (ctx->resident_limit > 0) {
        const int64_t stride = ctx->cfg.stride;
        for (int64_t i = 0; i < n; i++) {
            const int64_t len = out_offsets[i + 1] - out_offsets[i];
            if (len > 0 && (first[i] < 0 || (first[i] + len) * stride > ctx->resident_limit))
                return fail(ctx, PXG_E_STATE, "pxg_batch_pooled_signal: a stretch ends behind the prefix this batch was "
                                              "staged with (pxg_batch_stage_prefix); stage whole reads for it");
        }
    }
    int64_t* d_first = S.put(first, (size_t)n, ctx->stream);
    int64_t* d_ooff = S.put(out_offsets, (size_t)n + 1, ctx->stream);
    float* d_out = S.alloc<float>(total);
    int* d_flag = S.alloc<int>(1);
    HOOK_CHECK(d_first && d_ooff && d_out && d_flag);
    PXG_HIP(ctx, hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream));
    int rc = pxg_launch_pooled_stretch(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p, ctx->ss.p, d_first, d_ooff,
                                       d_out, d_flag);
    if (rc) return rc;
    int flag = 0;
    HOOK_GET(out, d_out, total);
    HOOK_GET(&flag, d_flag, 1);
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PXG_HIP(ctx, hipGetLastError());
    if (flag) return fail(ctx, PXG_E_INVALID, "pxg_batch_pooled_signal: a stretch leaves its read");
    return PXG_OK;
}

extern "C" int pxg_viterbi(pxg_ctx* ctx, int which_model, int64_t n, const float* signal_arena,
                           const int64_t* off, int32_t* seg_first, int32_t* seg_last,
                           int32_t* path_or_null, double* logp_or_null)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    if (which_model < 0 || which_model > 1) return fail(ctx, PXG_E_INVALID, "which_model");
    int64_t longest = 0;
    for (int64_t i = 0; i < n; i++) longest = std::max(longest, off[i + 1] - off[i]);
    if (longest >= (1 << 30)) return fail(ctx, PXG_E_UNSUPPORTED, "pxg_viterbi: sequences must be < 2^30 steps");
    float* d_sig = S.put(signal_arena, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    int32_t* d_segs = S.alloc<int32_t>((size_t)n * 2 * PXG_N_SEGMENTS);
    double* d_logp = S.alloc<double>((size_t)n);
    HOOK_CHECK(d_sig && d_off && d_segs && d_logp);
    int rc = pxg_launch_viterbi_f32(ctx, which_model, n, d_sig, d_off, (int)longest, d_segs, d_logp);
    if (rc) return rc;
    std::vector<int32_t> segs((size_t)n * 2 * PXG_N_SEGMENTS);
    HOOK_GET(segs.data(), d_segs, segs.size());
    if (logp_or_null) HOOK_GET(logp_or_null, d_logp, n);
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; i++) {
        const int32_t* f = &segs[(size_t)i * 2 * PXG_N_SEGMENTS];
        const int32_t* l = f + PXG_N_SEGMENTS;
        for (int s = 0; s < PXG_N_SEGMENTS; s++) {
            seg_first[i * PXG_N_SEGMENTS + s] = f[s];
            seg_last[i * PXG_N_SEGMENTS + s] = l[s];
            if (path_or_null && f[s] >= 0)     // left-to-right: runs ARE the path
                for (int t = f[s]; t <= l[s]; t++) path_or_null[off[i] + t] = s;
        }
    }
    HOOK_END
}

extern "C" int pxg_barcode_window(pxg_ctx* ctx, int64_t n, const float* signal_arena,
                                  const int64_t* off, float* out, int8_t* pushed)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    const size_t trim = (size_t)ctx->cfg.signal_trim_length;
    float* d_sig = S.put(signal_arena, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    float* d_out = S.alloc<float>((size_t)n * trim);
    int8_t* d_push = S.alloc<int8_t>((size_t)n);
    HOOK_CHECK(d_sig && d_off && d_out && d_push);
    PXG_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t)n * trim * sizeof(float), ctx->stream));
    int rc = pxg_launch_barcode_window_f32(ctx, n, d_sig, d_off, d_out, d_push);
    if (rc) return rc;
    HOOK_GET(out, d_out, (size_t)n * trim);
    HOOK_GET(pushed, d_push, n);
    HOOK_END
}

extern "C" int pxg_demux_lstm(pxg_ctx* ctx, int64_t n, const float* win, float* probs)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    const size_t trim = (size_t)ctx->cfg.signal_trim_length;
    const int C = ctx->cfg.demux_dense.out_dim;
    float* d_win = S.put(win, (size_t)n * trim, ctx->stream);
    float* d_bidir = S.alloc<float>((size_t)((n + 15) / 16 * 16) * trim * 96);
    float* d_probs = S.alloc<float>((size_t)n * PXG_MAX_CLASSES);
    HOOK_CHECK(d_win && d_bidir && d_probs);
    int rc = pxg_launch_demux_lstm(ctx, n, nullptr, nullptr, d_win, d_bidir, d_probs,
                                   PXG_T_DEMUX_BIDIR, PXG_T_DEMUX_TOP);
    if (rc) return rc;
    std::vector<float> tmp((size_t)n * PXG_MAX_CLASSES);
    HOOK_GET(tmp.data(), d_probs, tmp.size());
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < C; j++) probs[i * C + j] = tmp[(size_t)i * PXG_MAX_CLASSES + j];
    HOOK_END
}

extern "C" int pxg_detect_events(pxg_ctx* ctx, int64_t n, const float* signal_arena,
                                 const int64_t* off, int64_t cap, pxg_event* events,
                                 int64_t* n_events)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    if (cap < 1) return fail(ctx, PXG_E_INVALID, "max_events_per_window must be >= 1");
    struct EvRec { uint32_t start; float length, mean, stdv; };
    const size_t blocks = (size_t)(n + PXG_PA_LANES - 1) / PXG_PA_LANES;
    float* d_sig = S.put(signal_arena, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    EvRec* d_ev = S.alloc<EvRec>(blocks * (size_t)cap * PXG_PA_LANES);
    int64_t* d_cnt = S.alloc<int64_t>((size_t)n);
    HOOK_CHECK(d_sig && d_off && d_ev && d_cnt);
    int rc = pxg_launch_detect_events(ctx, n, d_sig, d_off, cap, d_ev, d_cnt);
    if (rc) return rc;
    std::vector<EvRec> ev(blocks * (size_t)cap * PXG_PA_LANES);
    HOOK_GET(ev.data(), d_ev, ev.size());
    HOOK_GET(n_events, d_cnt, n);
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t r = 0; r < n; r++) {
        const size_t blk = (size_t)r / PXG_PA_LANES, lane = (size_t)r % PXG_PA_LANES;
        const int64_t m = std::min<int64_t>(n_events[r], cap);
        for (int64_t q = 0; q < m; q++) {
            const EvRec& e = ev[(blk * (size_t)cap + (size_t)q) * PXG_PA_LANES + lane];
            pxg_event& o = events[r * cap + q];
            o.start = e.start; o.length = e.length; o.mean = e.mean; o.stdv = e.stdv;
            o.pos = -1; o.state = -1;               // csupport.c:156-159 defaults
        }
    }
    HOOK_END
}

extern "C" int pxg_guppy_event_means(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                                     const pxg_calib* calib, const float* scale_shift,
                                     const int64_t* first_sample, const int64_t* events_offsets,
                                     int32_t block_stride, float* mean, float* scaled_mean)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    for (int64_t i = 0; i < n; i++)
        if (first_sample[i] < 0 || events_offsets[i + 1] < events_offsets[i])
            return fail(ctx, PXG_E_INVALID, "pxg_guppy_event_means: bad first_sample / offsets");
    const size_t ne = (size_t)events_offsets[n];
    int16_t* d_raw = S.put(raw, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    pxg_calib* d_cal = S.put(calib, (size_t)n, ctx->stream);
    float* d_ss = S.put(scale_shift, (size_t)n * 2, ctx->stream);
    int64_t* d_first = S.put(first_sample, (size_t)n, ctx->stream);
    int64_t* d_eoff = S.put(events_offsets, (size_t)n + 1, ctx->stream);
    float* d_mean = S.alloc<float>(ne);
    float* d_scaled = S.alloc<float>(ne);
    HOOK_CHECK(d_raw && d_off && d_cal && d_ss && d_first && d_eoff && d_mean && d_scaled);
    int rc = pxg_launch_guppy_event_means(ctx, n, d_raw, d_off, d_cal, d_ss, d_first, d_eoff,
                                          block_stride, d_mean, d_scaled);
    if (rc) return rc;
    if (ne) {
        HOOK_GET(mean, d_mean, ne);
        HOOK_GET(scaled_mean, d_scaled, ne);
    }
    HOOK_END
}

extern "C" int pxg_batch_event_table(pxg_ctx* ctx, const int64_t* first_sample, const int64_t* events_offsets,
                                     int32_t block_stride, float* mean, float* stdv, float* scaled_mean)
{
    HOOK_BEGIN
    const int64_t n = ctx->n_reads;
    if (n <= 0) return fail(ctx, PXG_E_STATE, "pxg_batch_event_table: no resident batch");
    if (need_whole_reads(ctx, "pxg_batch_event_table")) return PXG_E_STATE;
    if (!(ctx->last_stage_mask & (PXG_STAGE_SCALER | PXG_STAGE_SEGMENT)))
        return fail(ctx, PXG_E_STATE, "pxg_batch_event_table: nothing has been run on the resident batch");
    if (!first_sample || !events_offsets || events_offsets[0] != 0)
        return fail(ctx, PXG_E_INVALID, "pxg_batch_event_table: bad arguments");
    for (int64_t i = 0; i < n; i++)
        if (events_offsets[i + 1] < events_offsets[i] || (events_offsets[i + 1] > events_offsets[i] && first_sample[i] < 0))
            return fail(ctx, PXG_E_INVALID, "pxg_batch_event_table: bad first_sample / offsets");
    const size_t ne = (size_t)events_offsets[n];
    if (!ne) return PXG_OK;
    if (!mean || !stdv || !scaled_mean) return fail(ctx, PXG_E_INVALID, "pxg_batch_event_table: an output is null");
    int64_t* d_first = S.put(first_sample, (size_t)n, ctx->stream);
    int64_t* d_eoff = S.put(events_offsets, (size_t)n + 1, ctx->stream);
    float* d_mean = S.alloc<float>(ne);
    float* d_sd = S.alloc<float>(ne);
    float* d_scaled = S.alloc<float>(ne);
    HOOK_CHECK(d_first && d_eoff && d_mean && d_sd && d_scaled);
    int rc = pxg_launch_guppy_event_means(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p, ctx->ss.p, d_first, d_eoff,
                                          block_stride, d_mean, d_scaled, d_sd);
    if (rc) return rc;
    HOOK_GET(mean, d_mean, ne);
    HOOK_GET(stdv, d_sd, ne);
    HOOK_GET(scaled_mean, d_scaled, ne);
    HOOK_END
}

extern "C" int pxg_batch_unsplit_scan(pxg_ctx* ctx, const int64_t* first_sample,
                                      const int64_t* n_blocks, int32_t block_stride,
                                      int64_t cap_intervals, int64_t* out_intervals,
                                      int32_t* out_count, int64_t* out_total)
{
    HOOK_BEGIN
    const int64_t n = ctx->n_reads;
    if (out_total) *out_total = 0;
    if (n <= 0) return PXG_OK;
    if (need_whole_reads(ctx, "pxg_batch_unsplit_scan")) return PXG_E_STATE;
    if (!first_sample || !n_blocks || !out_count || !out_total || cap_intervals < 0 ||
        (cap_intervals > 0 && !out_intervals))
        return fail(ctx, PXG_E_INVALID, "pxg_batch_unsplit_scan: bad arguments");
    if (block_stride < 1) return fail(ctx, PXG_E_INVALID, "pxg_batch_unsplit_scan: block_stride < 1");
    const pxg_config& c = ctx->cfg;
    const pxg_hmm& U = c.unsplit_model;
    if (U.n_states < 1 || U.adapter_state < 0 || U.leader_low_state < 0 || U.leader_high_state < 0)
        return fail(ctx, PXG_E_INVALID, "unsplit model lacks adapter / leader states");
    // window geometry bounds from the config and the batch's sampling-rate range: the largest
    // window (steps of back pointers a wave must hold) and the most windows a read can have
    const int64_t win_max = (int64_t)(c.unsplit_window_size * ctx->rate_max);
    const int64_t step_min = (int64_t)(c.unsplit_window_step * ctx->rate_min);
    if (step_min < 1 || win_max < 0 || win_max / block_stride + 2 > (1 << 22))
        return fail(ctx, PXG_E_INVALID, "unsplit_read_detection window_size / window_step out of range");
    const int tmax = (int)(win_max / block_stride + 2);
    const int wcand = pxg_unsplit_cand_slots(ctx, tmax, block_stride);
    // per-read frames; a read with an impossible frame gets an empty one and its own error code
    std::vector<int64_t> eoff((size_t)n + 1, 0), first((size_t)n);
    std::vector<int64_t> bad;
    int64_t units_bound = 0;
    for (int64_t i = 0; i < n; i++) {
        const bool ok = first_sample[i] >= 0 && n_blocks[i] >= 0;
        if (!ok) bad.push_back(i);
        first[i] = ok ? first_sample[i] : 0;
        const int64_t nb = ok ? n_blocks[i] : 0;
        eoff[i + 1] = eoff[i] + nb;
        units_bound += nb * block_stride / step_min + 2;
    }
    const size_t ne = (size_t)eoff[n];
    int rc;
    if ((rc = pxg_reserve(ctx, ctx->ev_first, (size_t)n)) || (rc = pxg_reserve(ctx, ctx->ev_off, (size_t)n + 1)) ||
        (rc = pxg_reserve(ctx, ctx->ev_mean, ne)) || (rc = pxg_reserve(ctx, ctx->ev_scaled, ne)) ||
        (rc = pxg_reserve(ctx, ctx->unit_off, (size_t)n + 1)) || (rc = pxg_reserve(ctx, ctx->n_win, (size_t)n)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_ivoff, (size_t)n + 1)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_iv, (size_t)std::max<int64_t>(cap_intervals, 1) * 2)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_cnt, (size_t)n)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_scr, pxg_unsplit_scratch_bytes(ctx, units_bound, tmax))) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_cand, pxg_unsplit_cand_bytes(units_bound, wcand))))
        return rc;
    int64_t* d_first = ctx->ev_first.p;
    int64_t* d_eoff = ctx->ev_off.p;
    ScanStreamScope scan(ctx);          // beside K6 when the last run had the poly(A) stage
    // (through the context's page-locked mirrors, like every other small host array of a batch: the runtime is never
    //  handed pageable memory whose lifetime ends with this call -- ADVICE r4, profiles/r05/fault_hunt.md)
    if ((rc = pxg_h2d_meta(ctx, 2, 0, d_first, first.data(), (size_t)n * sizeof(int64_t), ctx->stream)) ||
        (rc = pxg_h2d_meta(ctx, 2, 1, d_eoff, eoff.data(), ((size_t)n + 1) * sizeof(int64_t), ctx->stream)))
        return rc;
    StreamSyncOnError guard(ctx->stream);
    pxg_timer_begin(ctx, PXG_T_EVENT_MEANS);
    rc = pxg_launch_guppy_event_means(ctx, n, ctx->raw.p, ctx->offsets.p, ctx->calib.p, ctx->ss.p,
                                      d_first, d_eoff, block_stride, ctx->ev_mean.p, ctx->ev_scaled.p);
    if (rc) return rc;
    pxg_timer_end(ctx, PXG_T_EVENT_MEANS);
    if (scan.gate_polya()) return fail(ctx, PXG_E_HIP, "pxg_batch_unsplit_scan: scan stream gate");
    pxg_timer_begin(ctx, PXG_T_UNSPLIT);
    // plan -> unit offsets -> scan -> per-read counts -> interval offsets -> compact gather,
    // all on the stream: nothing here waits for the device
    if ((rc = pxg_launch_unsplit_plan(ctx, n, ctx->calib.p, ctx->status.p, ctx->segs.p, d_first, d_eoff,
                                      block_stride, ctx->n_win.p)) ||
        (rc = pxg_launch_exclusive_scan(ctx, n, ctx->n_win.p, ctx->unit_off.p)) ||
        (rc = pxg_launch_unsplit_scan(ctx, n, units_bound, tmax, ctx->calib.p, ctx->status.p, ctx->segs.p,
                                      d_first, d_eoff, ctx->unit_off.p, ctx->ev_scaled.p, block_stride,
                                      ctx->unsplit_scr.p, ctx->unsplit_cand.p, wcand, ctx->unsplit_cnt.p)) ||
        (rc = pxg_launch_exclusive_scan(ctx, n, ctx->unsplit_cnt.p, ctx->unsplit_ivoff.p)) ||
        (rc = pxg_launch_unsplit_gather(ctx, n, units_bound, ctx->unit_off.p, ctx->unsplit_cand.p, wcand,
                                        ctx->unsplit_cnt.p, ctx->unsplit_ivoff.p, cap_intervals,
                                        ctx->unsplit_iv.p)))
        return rc;
    pxg_timer_end(ctx, PXG_T_UNSPLIT);
    if (scan.join()) return fail(ctx, PXG_E_HIP, "pxg_batch_unsplit_scan: joining the scan stream");
    HOOK_GET(out_count, ctx->unsplit_cnt.p, n);
    HOOK_GET(out_total, ctx->unsplit_ivoff.p + n, 1);
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    guard.dismiss();
    const int64_t got = std::min(*out_total, cap_intervals);
    if (got > 0) HOOK_GET(out_intervals, ctx->unsplit_iv.p, (size_t)got * 2);
    for (int64_t i : bad) out_count[i] = PXG_UNSPLIT_E_GEOMETRY;
    HOOK_END
}

// a19 for reads whose basecall brings its own event table (albacore's 14-column `Events', which
// fast5_file.py:178-179 passes through unchanged): the windows are the events with left <= start <= left +
// window (signal_analyzer.py:384-386) of the table's own ascending `start' column, an event ends where the
// next one starts (:321-324), scaled_mean = fl(fl(scale * mean) + shift) of the table's float32 `mean'
// column (:318).  Same kernels as the Guppy block frame, with the starts read instead of computed.
extern "C" int pxg_batch_unsplit_scan_events(pxg_ctx* ctx, const int64_t* n_events, const int64_t* ev_start,
                                             const float* ev_mean, int64_t cap_intervals, int64_t* out_intervals,
                                             int32_t* out_count, int64_t* out_total)
{
    HOOK_BEGIN
    const int64_t n = ctx->n_reads;
    if (out_total) *out_total = 0;
    if (n <= 0) return PXG_OK;
    if (!n_events || !out_count || !out_total || cap_intervals < 0 || (cap_intervals > 0 && !out_intervals))
        return fail(ctx, PXG_E_INVALID, "pxg_batch_unsplit_scan_events: bad arguments");
    const pxg_config& c = ctx->cfg;
    const pxg_hmm& U = c.unsplit_model;
    if (U.n_states < 1 || U.adapter_state < 0 || U.leader_low_state < 0 || U.leader_high_state < 0)
        return fail(ctx, PXG_E_INVALID, "unsplit model lacks adapter / leader states");
    const int64_t win_max = (int64_t)(c.unsplit_window_size * ctx->rate_max);
    const int64_t step_min = (int64_t)(c.unsplit_window_step * ctx->rate_min);
    if (step_min < 1 || win_max < 0) return fail(ctx, PXG_E_INVALID, "unsplit_read_detection window_size / window_step out of range");
    // per-read tables; the most events any window [left, left + win_max] can hold bounds the back-pointer rows
    std::vector<int64_t> eoff((size_t)n + 1, 0);
    std::vector<int64_t> bad;
    int64_t units_bound = 0, tmax64 = 2;
    for (int64_t i = 0; i < n; i++) {
        const int64_t ne = n_events[i] > 0 ? n_events[i] : 0;
        eoff[i + 1] = eoff[i] + ne;
        if (n_events[i] < 0) bad.push_back(i);
    }
    if (eoff[n] > 0 && (!ev_start || !ev_mean)) return fail(ctx, PXG_E_INVALID, "pxg_batch_unsplit_scan_events: null event columns");
    for (int64_t i = 0; i < n; i++) {
        const int64_t ne = eoff[i + 1] - eoff[i];
        if (!ne) continue;
        const int64_t* st = ev_start + eoff[i];
        bool sorted = st[0] >= 0;
        for (int64_t k = 1; k < ne && sorted; k++) sorted = st[k] >= st[k - 1];
        if (!sorted) return fail(ctx, PXG_E_INVALID, "pxg_batch_unsplit_scan_events: `start' column is not ascending");
        units_bound += (st[ne - 1] + 1) / step_min + 2;
        for (int64_t a = 0, b = 0; a < ne; a++) {          // events within win_max of event a
            while (b < ne && st[b] <= st[a] + win_max) b++;
            tmax64 = std::max(tmax64, b - a + 1);
        }
    }
    if (tmax64 > (1 << 22)) return fail(ctx, PXG_E_INVALID, "event table too dense for the scan window");
    const int tmax = (int)tmax64;
    const double shortest = std::min(c.unsplit_loosen_full_length, c.unsplit_strict_full_length);
    const int64_t min_cut = (int64_t)(shortest * ctx->rate_min);
    const int64_t by_blocks = (int64_t)tmax / 2 + 2, by_length = min_cut > 0 ? (win_max + 1) / min_cut + 2 : by_blocks;
    const int wcand = (int)std::max<int64_t>(1, std::min(by_blocks, by_length));
    const size_t ne_all = (size_t)eoff[n];
    int rc;
    if ((rc = pxg_reserve(ctx, ctx->ev_first, (size_t)n)) || (rc = pxg_reserve(ctx, ctx->ev_off, (size_t)n + 1)) ||
        (rc = pxg_reserve(ctx, ctx->ev_tstart, ne_all)) ||
        (rc = pxg_reserve(ctx, ctx->ev_mean, ne_all)) || (rc = pxg_reserve(ctx, ctx->ev_scaled, ne_all)) ||
        (rc = pxg_reserve(ctx, ctx->unit_off, (size_t)n + 1)) || (rc = pxg_reserve(ctx, ctx->n_win, (size_t)n)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_ivoff, (size_t)n + 1)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_iv, (size_t)std::max<int64_t>(cap_intervals, 1) * 2)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_cnt, (size_t)n)) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_scr, pxg_unsplit_scratch_bytes(ctx, units_bound, tmax))) ||
        (rc = pxg_reserve(ctx, ctx->unsplit_cand, pxg_unsplit_cand_bytes(units_bound, wcand))))
        return rc;
    PXG_HIP(ctx, hipMemsetAsync(ctx->ev_first.p, 0, (size_t)n * sizeof(int64_t), ctx->stream));
    StreamSyncOnError guard(ctx->stream);       // (from the first copy that may read the caller's arrays asynchronously)
    if ((rc = pxg_h2d_meta(ctx, 2, 1, ctx->ev_off.p, eoff.data(), ((size_t)n + 1) * sizeof(int64_t), ctx->stream)) ||
        // the per-EVENT columns (8 + 4 bytes x ~4 000 events per read) go in bounded page-locked chunks, not through a
        // mirror of their own size (pxg_h2d_meta is for the O(n_reads) arrays)
        (ne_all && ((rc = pxg_h2d_big(ctx, ctx->ev_tstart.p, ev_start, ne_all * sizeof(int64_t), ctx->stream)) ||
                    (rc = pxg_h2d_big(ctx, ctx->ev_mean.p, ev_mean, ne_all * sizeof(float), ctx->stream)))))
        return rc;
    pxg_timer_begin(ctx, PXG_T_EVENT_MEANS);
    if ((rc = pxg_launch_scale_event_means(ctx, n, (int64_t)ne_all, ctx->ev_off.p, ctx->ss.p, ctx->ev_mean.p, ctx->ev_scaled.p)))
        return rc;
    pxg_timer_end(ctx, PXG_T_EVENT_MEANS);
    pxg_timer_begin(ctx, PXG_T_UNSPLIT);
    if ((rc = pxg_launch_unsplit_plan(ctx, n, ctx->calib.p, ctx->status.p, ctx->segs.p, ctx->ev_first.p, ctx->ev_off.p, 1,
                                      ctx->n_win.p, ctx->ev_tstart.p)) ||
        (rc = pxg_launch_exclusive_scan(ctx, n, ctx->n_win.p, ctx->unit_off.p)) ||
        (rc = pxg_launch_unsplit_scan(ctx, n, units_bound, tmax, ctx->calib.p, ctx->status.p, ctx->segs.p, ctx->ev_first.p,
                                      ctx->ev_off.p, ctx->unit_off.p, ctx->ev_scaled.p, 1, ctx->unsplit_scr.p,
                                      ctx->unsplit_cand.p, wcand, ctx->unsplit_cnt.p, ctx->ev_tstart.p)) ||
        (rc = pxg_launch_exclusive_scan(ctx, n, ctx->unsplit_cnt.p, ctx->unsplit_ivoff.p)) ||
        (rc = pxg_launch_unsplit_gather(ctx, n, units_bound, ctx->unit_off.p, ctx->unsplit_cand.p, wcand,
                                        ctx->unsplit_cnt.p, ctx->unsplit_ivoff.p, cap_intervals, ctx->unsplit_iv.p)))
        return rc;
    pxg_timer_end(ctx, PXG_T_UNSPLIT);
    HOOK_GET(out_count, ctx->unsplit_cnt.p, n);
    HOOK_GET(out_total, ctx->unsplit_ivoff.p + n, 1);
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    guard.dismiss();
    const int64_t got = std::min(*out_total, cap_intervals);
    if (got > 0) HOOK_GET(out_intervals, ctx->unsplit_iv.p, (size_t)got * 2);
    for (int64_t i : bad) out_count[i] = PXG_UNSPLIT_E_GEOMETRY;
    HOOK_END
}

// The two mutexes of pxg_process_batch_ex for a caller that drives the split calls itself (a batch with
// dump options, several Guppy block strides or albacore tables takes stage / swap / run / scans / downloads
// one by one): which = 0 the spare input slot, 1 the resident batch; take 0, then 1, release 0 after the
// swap and 1 after the last download -- the order the one-call form uses, so both forms may be mixed
// from any number of threads.
extern "C" int pxg_ctx_lock(pxg_ctx* ctx, int which)
{
    if (!ctx || which < 0 || which > 1) return PXG_E_INVALID;
    (which ? ctx->mt_run : ctx->mt_stage).lock();
    return PXG_OK;
}

extern "C" int pxg_ctx_unlock(pxg_ctx* ctx, int which)
{
    if (!ctx || which < 0 || which > 1) return PXG_E_INVALID;
    (which ? ctx->mt_run : ctx->mt_stage).unlock();
    return PXG_OK;
}

extern "C" int pxg_polya(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                         const pxg_calib* calib, const float* scale_shift, const int32_t* seg_first,
                         const int32_t* seg_last, pxg_read_result* out, int64_t spike_cap,
                         pxg_polya_spike* spikes_or_null, int64_t* spike_offsets_or_null)
{
    HOOK_BEGIN
    if (n <= 0) return PXG_OK;
    if (!off || !calib || !scale_shift || !seg_first || !seg_last || !out || (off[n] > 0 && !raw) ||
        (spikes_or_null && !spike_offsets_or_null))
        return fail(ctx, PXG_E_INVALID, "pxg_polya: bad arguments");
    // the hook shares the overflow list and the event scratch with the resident batch: a batch
    // whose retries are still pending (run, not yet downloaded) is settled first
    int rc = settle_polya(ctx);
    if (rc) return rc;
    std::vector<int32_t> segs((size_t)n * 2 * PXG_N_SEGMENTS);
    for (int64_t i = 0; i < n; i++)
        for (int q = 0; q < PXG_N_SEGMENTS; q++) {
            segs[(size_t)i * 2 * PXG_N_SEGMENTS + q] = seg_first[i * PXG_N_SEGMENTS + q];
            segs[(size_t)i * 2 * PXG_N_SEGMENTS + PXG_N_SEGMENTS + q] = seg_last[i * PXG_N_SEGMENTS + q];
        }
    int16_t* d_raw = S.put(raw, (size_t)off[n], ctx->stream);
    int64_t* d_off = S.put(off, (size_t)n + 1, ctx->stream);
    pxg_calib* d_cal = S.put(calib, (size_t)n, ctx->stream);
    float* d_ss = S.put(scale_shift, (size_t)n * 2, ctx->stream);
    int32_t* d_segs = S.put(segs.data(), segs.size(), ctx->stream);
    int32_t* d_status = S.alloc<int32_t>((size_t)n);
    int32_t* d_pout = S.alloc<int32_t>((size_t)n * 8);
    HOOK_CHECK(d_raw && d_off && d_cal && d_ss && d_segs && d_status && d_pout);
    struct Arena {                      // the hook's own spike arena, gone with the call
        DevBuf<pxg_polya_spike> b;
        ~Arena() { if (b.p) (void)hipFree(b.p); }
    } arena;
    PXG_HIP(ctx, hipMemsetAsync(d_status, 0, (size_t)n * sizeof(int32_t), ctx->stream));
    if ((rc = pxg_launch_polya(ctx, n, d_raw, d_off, d_cal, d_ss, d_status, d_segs, d_pout, arena.b))) return rc;
    int64_t rows = 0;
    if ((rc = pxg_polya_settle(ctx, n, longest_of(off, n), d_raw, d_off, d_cal, d_ss, d_status, d_segs, d_pout,
                               arena.b, nullptr, &rows)))
        return rc;
    std::vector<int32_t> po((size_t)n * 8);
    HOOK_GET(po.data(), d_pout, po.size());
    PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, (size_t)n * sizeof(pxg_read_result));
    for (int64_t i = 0; i < n; i++) {
        pxg_read_result& o = out[i];
        const int32_t* q = &po[(size_t)i * 8];
        o.status = PXG_ST_OKAY;
        o.n_pooled = (int32_t)((off[i + 1] - off[i]) / ctx->cfg.stride);
        for (int k = 0; k < PXG_N_SEGMENTS; k++) {
            o.seg_first[k] = seg_first[i * PXG_N_SEGMENTS + k];
            o.seg_last[k] = seg_last[i * PXG_N_SEGMENTS + k];
        }
        o.scale = scale_shift[2 * i];
        o.shift = scale_shift[2 * i + 1];
        o.bc_label = -1;
        o.polya_called = (int8_t)q[0];
        o.polya_n_spikes = q[1];
        o.polya_dwell_samples = q[2];
        o.polya_begin = (int64_t)(((uint64_t)(uint32_t)q[4] << 32) | (uint32_t)q[3]);
        o.polya_end = (int64_t)(((uint64_t)(uint32_t)q[6] << 32) | (uint32_t)q[5]);
    }
    if (spike_offsets_or_null &&
        (rc = pxg_polya_collect_spikes(ctx, n, d_pout, arena.b, rows, spike_cap, spikes_or_null, spike_offsets_or_null)))
        return rc;
    HOOK_END
}
