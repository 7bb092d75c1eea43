// ResShift engine: model graphs (UNetModelSwin, VQModelTorch), weight packing, scratch arena, the
// sampling loop and the C ABI declared in include/resshift_hip.h.
//
// Reference behaviour reproduced here (file:line into the reference repo):
//   UNetModelSwin construction / forward      models/unet.py:632-895
//   ResBlock                                  models/unet.py:110-206
//   BasicLayer / SwinTransformerBlock         models/swin_transformer.py:163-281,348-442
//   VQModelTorch encode/decode                ldm/models/autoencoder.py:28-40
//   Encoder / Decoder / ResnetBlock / AttnBlock  ldm/modules/diffusionmodules/model.py:90-203,452-660
//   p_sample_loop                             models/gaussian_diffusion.py:367-529
//
// Design notes
//   * NHWC activations; every conv/linear is one implicit-GEMM launch with fused bias/GELU/residual.
//   * Skip concatenations (unet.py:891) are zero-copy: each input block writes its output straight
//     into the upper channel slice of the buffer the matching output block will read, and the
//     decoder path writes into the lower slice (pixel stride = total channels).
//   * FiLM vectors (time_embed + every ResBlock's emb_layers) depend only on t: computed once per
//     distinct t on device and cached.
//   * All weights live in ONE caller-owned device blob whose layout is a pure function of the
//     config, so a multi-GPU host can broadcast it with a single RCCL call.
#include "common.h"
#include "../../include/resshift_hip.h"
#include <dlfcn.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" {
int rs_igemm_launch(const IGemmParams* p, int in_dt, int out_dt, int nz, hipStream_t st);
int rs_igemm_splitk_plan(int M, int Cout, int Ktot, int in_dt);
int rs_igemm4_pick(const IGemmParams* p, int in_dt, int out_dt, int nz, int* TW, int* BC);
int rs_igemm4_plan(const IGemmParams* p, int in_dt, int out_dt, int nz, int* TW, int* BC, int* SEG, int* SK);
int rs_igemm4_stats_px(const IGemmParams* p, int in_dt);
int rs_wino_plan(const IGemmParams* p, int in_dt, int out_dt, int nz);
int rs_wino_launch(const IGemmParams* p, hipStream_t st);
size_t rs_wino_weight_bytes(int Cin, int Cout);
float rs_wino_pack(const float* w_ref, int Cin, int Cout, void* dst);
int rs_wino_stats_px();
int rs_wino_tiles(const IGemmParams* p);
int rs_igemm_split_stats_px(const IGemmParams* p, int splitk);
int rs_direct_conv_launch(const DirectConvParams* p, int in_dt, int out_dt, hipStream_t st);
int rs_head_conv_launch(const void* x, int in_dt, const float* coef_dev, const float* w_dev, const float* bias_dev, float* y, int B, int H, int W, int C,
                        int ldx, int Cout, int ldy, hipStream_t st);
int rs_groupnorm_launch(const GNParams* p, int dt, int apply_slabs, hipStream_t st);
int rs_win_attn_launch(const WinAttnParams* p, int dt, hipStream_t st);
int rs_softmax_rows_launch(const float* s, void* out, int out_dt, long long nrows, int ncols, long long lds_, long long ldo, hipStream_t st);
int rs_nchw_to_nhwc_launch(const float* in, void* out, int out_dt, int B, int C, int HW, int ldo, int coff, float scale, hipStream_t st);
int rs_nhwc_to_nchw_launch(const void* in, int in_dt, float* out, int B, int C, int HW, int ldi, int coff, hipStream_t st);
int rs_axpbypcz_launch(const float* x, const float* z, const float* n, float* y, float a, float b, float c, long long cnt, hipStream_t st);
int rs_clamp_launch(float* x, float lo, float hi, long long cnt, hipStream_t st);
int rs_win_attn_qkv_supported(int heads, int E);
int rs_win_attn_qkv_launch(const WinAttnParams* p, hipStream_t st);
int rs_win_attn_qkv_split_launch(const WinAttnParams* p, hipStream_t st);
int rs_ae_flash_supported(int C, int T);
int rs_ae_flash_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, const float* bv, void* o, int ldo, int nz, int T, int C,
                       float scale, hipStream_t st);
int rs_ae_flash_split_supported(int C, int T);
int rs_ae_flash_split_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, const float* bv, void* o, int ldo, int nz, int T, int C,
                             float scale, hipStream_t st);
int rs_swin_mlp_supported(int E, int HD);
int rs_swin_mlp_split_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y, int M, int ldx,
                             int ldres, int ldy, int E, int HD, const float* xcoef, int HW, float* ystats, int ystats_ld, const GNTail* tail, hipStream_t st);
int rs_swin_mlp_split_launch_n(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y, int M, int ldx,
                               int ldres, int ldy, int E, int HD, int NO, const float* xcoef, int HW, float* ystats, int ystats_ld, const GNTail* tail,
                               hipStream_t st);
int rs_swin_mlp_split_unembed_supported(int E, int HD, int NO);
int rs_swin_mlp_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y, int M, int ldx,
                       int ldres, int ldy, int E, int HD, const float* xcoef, int HW, float* ystats, int ystats_ld, hipStream_t st);
int rs_small_linear_launch(const float* x, const float* w, const float* bias, float* y, int R, int K, int N, int silu_in, int silu_out, hipStream_t st);
int rs_bicubic_launch(const float* in, void* out, int out_dt, int B, int C, int H, int W, int sf, int ldo, hipStream_t st);
int rs_vq_launch(const float* z, const float* codebook, float* zq, int* idx, long long N, int NE, int D, hipStream_t st);
int rs_copy_channels_launch(const void* src, int lds_, void* dst, int ldd, int C, long long npix, int dt, hipStream_t st);
int rs_convert_launch(const void* src, int src_dt, void* dst, int dst_dt, int C, long long npix, hipStream_t st);
}

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

// Fragment-major order of a [N][K] weight (ConvW::wh_frag / ws_frag): element e of lane (lr = lane & 15, lg = lane >> 4) of (16-row block
// nb, k step ks) is W[16 nb + lr][32 ks + 8 lg + e].  `hi(n, k)` / `lo(n, k)` fetch the fp16 planes; with a lo plane every (nb, ks) holds
// 1 KB of hi followed by 1 KB of lo.
template <class FH, class FL>
static void frag_major_fill(int N, int K, f16* dst, bool with_lo, FH hi, FL lo) {
    const int KS = K / 32, parts = with_lo ? 2 : 1;
    for (int nb = 0; nb < N / 16; ++nb)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * nb + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + e;
                    const size_t o = (((size_t)(nb * KS + ks) * parts) * 64 + lane) * 8 + e;
                    dst[o] = hi(n, k);
                    if (with_lo) dst[o + 512] = lo(n, k);
                }
}
// from the reference fp32 weight: fp16 copy (out16) or split (hi, lo) copy (outsplit), rounded exactly like add_conv's packers
static void rs_pack_frag_major(const float* w, int N, int K, f16* out16, f16* outsplit) {
    if (out16) frag_major_fill(N, K, out16, false, [&](int n, int k) { return (f16)w[(size_t)n * K + k]; }, [&](int, int) { return (f16)0.f; });
    if (outsplit)
        frag_major_fill(N, K, outsplit, true, [&](int n, int k) { f16 h, l; rs_split(w[(size_t)n * K + k], h, l); return h; },
                        [&](int n, int k) { f16 h, l; rs_split(w[(size_t)n * K + k], h, l); return l; });
}
// from a row-major DEVICE operand (op-level test entries): rows of `ld` halfs, lo plane at +lo_off halfs (< 0: fp16 only) -> device copy
static void* frag_major_from_device_rows(const void* wdev, int N, int K, int ld, int lo_off) {
    std::vector<f16> rows((size_t)N * ld), out((size_t)N * K * (lo_off >= 0 ? 2 : 1));
    if (hipMemcpy(rows.data(), wdev, rows.size() * sizeof(f16), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    frag_major_fill(N, K, out.data(), lo_off >= 0, [&](int n, int k) { return rows[(size_t)n * ld + k]; },
                    [&](int n, int k) { return rows[(size_t)n * ld + (lo_off >= 0 ? lo_off : 0) + k]; });
    void* d = nullptr;
    if (hipMalloc(&d, out.size() * sizeof(f16)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, out.data(), out.size() * sizeof(f16), hipMemcpyHostToDevice);
    return d;
}

struct View {
    void* p = nullptr; int B = 0, H = 0, W = 0, C = 0, ld = 0, dt = RS_F16;
    // optional per-channel partial statistics of this tensor, [B][stS][stld][2] floats (IGemmParams::ystats): set by whoever
    // allocates the tensor when its producer is the halo conv kernel; the consuming GroupNorm then skips its statistics pass
    float* st = nullptr; int stS = 0, stld = 0;
    int st_prod = -1;   // sequence number of the launch that produces `st` (Exec::prod_seq): key of the GroupNorm tail plan
    // a channel concatenation (models/unet.py:891) whose halves have different producers: `st` covers channels [0, st_n0), `st2` the
    // rest (column 0 of an st2 row = channel st_n0).  st2 == nullptr: `st` covers all C channels.
    float* st2 = nullptr; int st2S = 0, st2ld = 0, st_n0 = 0;
    bool stats_complete() const { return st != nullptr; }
    long long pixels() const { return (long long)B * H * W; }
    View slice(int c0, int c) const {
        View v = *this; v.p = (char*)p + (size_t)c0 * rs_dtype_chan_bytes(dt); v.C = c;
        if (st2) {   // a slice of a concatenation keeps statistics only when it is exactly one of the halves
            v.st2 = nullptr; v.st2S = v.st2ld = v.st_n0 = 0;
            if (c0 == 0 && c == st_n0) { /* first half: st as is */ }
            else if (c0 == st_n0 && c == C - st_n0) { v.st = st2; v.stS = st2S; v.stld = st2ld; v.st_prod = -1; }
            else { v.st = nullptr; v.stS = v.stld = 0; v.st_prod = -1; }
        } else if (st) v.st = st + 2 * (size_t)c0;
        return v;
    }
};

// ------------------------------------------------------------------ weight blob
struct Blob {
    char* base = nullptr;
    size_t off = 0;
    bool fill = false;
    std::vector<char> staging;
    void* add(size_t bytes, const std::function<void(char*)>& filler) {
        const size_t o = (off + 255) & ~(size_t)255;
        off = o + bytes;
        if (fill) filler(staging.data() + o);
        return base + o;  // only meaningful once bound
    }
};

struct ConvW {
    int Cin = 0, CinP = 0, Cout = 0, KH = 1, KW = 1;
    void* wh = nullptr; void* wf = nullptr; void* ws = nullptr; float* wd = nullptr; float* bias = nullptr;
    // fragment-major copies for the fused window-attention kernels (add_frag_copies): [16-row block][32-wide k step][lane] x 16 B (fp16;
    // split: 1 KB of hi then 1 KB of lo per block and k step), so that a wave's A-operand fragment is ONE contiguous 1 KB read
    void* wh_frag = nullptr; void* ws_frag = nullptr;
    void* ww = nullptr;   // Winograd F(2x2,3x3) form of a 3x3 conv's split weights (wino.hip: rs_wino_pack order), packed only with RS_WINO=1
    bool direct = false;
    int idx = -1;   // position in rs_engine::big_w (per-layer "|w| >= 30" flags, see there)
    const void* w_for(int dt) const { return dt == RS_F16 ? wh : (dt == RS_F16S ? ws : wf); }
    const void* w_frag_for(int dt) const { return dt == RS_F16 ? wh_frag : (dt == RS_F16S ? ws_frag : nullptr); }
};
struct GNW { float* gamma = nullptr; float* beta = nullptr; int C = 0; };
struct ResBlockW { GNW n1, n2; ConvW c1, c2, skip; bool has_skip = false; int Cin = 0, Cout = 0; int film_off = -1; ConvW emb; };
struct SwinBlockW { GNW n1, n2; ConvW qkv, proj, fc1, fc2; float* bias_t = nullptr; float* bias_n = nullptr; float* bias_c = nullptr; int shift = 0; };
struct BasicLayerW { ConvW embed, unembed; std::vector<SwinBlockW> blocks; int C = 0, E = 0;
                     ConvW unfold; bool has_unfold = false; };   // unfold: [Wu W2 | Wu] of the last block's fc2 and patch_unembed (basiclayer())
struct UBlock {
    bool has_conv = false, has_res = false, has_swin = false, has_down = false, has_up = false;
    ConvW conv; ResBlockW res; BasicLayerW swin; int out_ch = 0; int level = 0;
    ConvW upf[4]; bool has_upf = false;   // sub-pixel form of the upsampling conv (add_upfold())
};
struct AttnW { GNW norm; ConvW q, k, v, proj; int C = 0; };
struct AELevel { std::vector<ResBlockW> blocks; bool has_resample = false; ConvW resample; ConvW upf[4]; bool has_upf = false; };

struct Arena {
    char* base = nullptr; size_t cap = 0, off = 0, peak = 0;
    void* alloc(size_t n) {
        off = (off + 255) & ~(size_t)255;
        void* p = base + off;
        off += n;
        peak = std::max(peak, off);
        return p;
    }
};

// GroupNorm tail plan (gn_tail.h): the launch that completes a tensor's statistics also writes the coefficients of the GroupNorm that
// consumes it.  The producer runs BEFORE the consumer is known, so the plan is made by the dry sizing pass - which walks the same
// control flow as the real pass - and keyed by the producer's sequence number: the consuming gn_coef() call of the dry pass claims its
// producer, the real pass attaches the tail at the producer's launch and skips the coefficient launch at the consumer.
struct TailPlan {
    bool on = false; int consumer = -1;
    const float* gamma = nullptr; const float* beta = nullptr; const float* film = nullptr; float eps = 0.f;
    int C = 0, HW = 0; size_t coef_off = 0;
    bool two = false; size_t st2_off = 0; int st2S = 0, st2ld = 0;   // the other half of a concatenation (statistics live in the pool)
    mutable bool drawn = false;   // (real pass: the producer's launch attached the tail - checked against the plan at the end of the call)
};

struct Exec {
    hipStream_t st = nullptr; Arena* arena = nullptr; bool dry = false; long long launches = 0; int err = 0;
    bool dbg = false;                        // debug trace requested (both passes): no statistics fusion, no tails
    std::vector<TailPlan>* plan = nullptr;   // by producer sequence number
    int prod_seq = 0, gn_seq = 0;
    // coefficient pool ([B][2][C] affines; behind the scratch arena, reset per network body - stream order keeps reuse safe) and the
    // ticket pool of the tails (zeroed once per call)
    char* pool_base = nullptr; size_t pool_off = 0, pool_peak = 0;
    unsigned* ticket_base = nullptr; size_t ticket_used = 0;
    float* pool(size_t bytes) {
        pool_off = (pool_off + 255) & ~(size_t)255;
        float* q = (float*)(pool_base + pool_off);
        pool_off += bytes;
        pool_peak = std::max(pool_peak, pool_off);
        return q;
    }
    unsigned* tickets(int n) { unsigned* q = ticket_base + ticket_used; ticket_used += (size_t)n; return q; }
    const TailPlan* tail_of(int prod) const { return (plan && prod >= 0 && prod < (int)plan->size() && (*plan)[prod].on) ? &(*plan)[prod] : nullptr; }
    // the GNTail block of a producer's launch from its plan entry (real pass): the consuming GroupNorm's parameters, its coefficient slot, B
    // fresh tickets and the other half of a concatenation; the launcher adds the arrival count and segment 0 (= this launch's statistics)
    bool fill_tail(int prod, int B, GNTail& g) {
        const TailPlan* t = tail_of(prod);
        if (!t) return false;
        t->drawn = true;
        g.gamma = t->gamma; g.beta = t->beta; g.film = t->film; g.eps = t->eps;
        g.coef = (float*)(pool_base + t->coef_off); g.ticket = tickets(B);
        g.C = t->C; g.groups = 32; g.HW = t->HW;
        if (t->two) { g.st1 = (const float*)(pool_base + t->st2_off); g.S1 = t->st2S; g.ld1 = t->st2ld; }
        return true;
    }
    bool used_split = false;   // a split-storage tensor was allocated: the call needs the split weights (checked after the dry run)
    View T(int B, int H, int W, int C, int dt) {
        if (dt == RS_F16S) used_split = true;
        View v; v.B = B; v.H = H; v.W = W; v.C = C; v.ld = C; v.dt = dt;
        v.p = arena->alloc((size_t)B * H * W * C * rs_dtype_size(dt));
        return v;
    }
    void* raw(size_t bytes) { return arena->alloc(bytes); }
    size_t mark() const { return arena->off; }
    bool keep = false;  // debug trace: never recycle scratch so that every recorded view stays valid
    std::vector<std::pair<std::string, View>>* trace = nullptr;
    std::string prefix;
    void reset(size_t m) { if (!keep) arena->off = m; }
    void tr(const std::string& name, const View& v) { if (trace && !dry) trace->emplace_back(prefix + name, v); }
    void check(int rc, const char* what) {
        ++launches;
        if (rc != 0 && err == 0) { err = rc; g_err = std::string("launch failed: ") + what; }
    }
    // MFMA implicit-GEMM launches are the roofline-relevant kernel family: count their algorithmic FLOPs
    // (2*M*N*K per batch entry, K = taps*Cin as in the usual conv FLOP count) and, when profiling is on, bracket
    // every launch with hipEvents on the launch stream.
    struct Prof { bool on = false; std::vector<hipEvent_t> ev; size_t used = 0; } * prof = nullptr, *prof_gn = nullptr;
    double gn_bytes = 0.0;               // algorithmic HBM bytes of the GroupNorm family: input read once + output written once
    long long gn_launches = 0;
    // event pair of one bracketed launch (null pair when profiling is off)
    static void bracket(Prof* pr, hipStream_t st, hipEvent_t& e0, hipEvent_t& e1) {
        e0 = e1 = nullptr;
        if (!pr || !pr->on) return;
        if (pr->used + 2 > pr->ev.size()) {
            const size_t old = pr->ev.size();
            pr->ev.resize(old + 1024);
            for (size_t i = old; i < pr->ev.size(); ++i) (void)hipEventCreate(&pr->ev[i]);
        }
        e0 = pr->ev[pr->used++]; e1 = pr->ev[pr->used++];
        (void)hipEventRecord(e0, st);
    }
    double igemm_flops[3] = {0.0, 0.0, 0.0};  // per input precision (fp16, fp32, split)
    double igemm_bytes = 0.0;            // algorithmic (compulsory) HBM bytes: source tensor + weights + output (+ residual), once each
    long long igemm_launches = 0;
    // per kernel family of the MFMA path (rs_profile_families): algorithmic FLOPs, launches, and the family of every bracket
    enum Fam { F_HALO16 = 0, F_HALO_SPLIT, F_IGEMM16, F_IGEMM_SPLIT, F_IGEMM32, F_WINATTN, F_SWINMLP, F_WINATTN_S, F_SWINMLP_S, F_AEFLASH, F_AEFLASH_S, F_WINO_S, F_COUNT };
    double fam_flops[F_COUNT] = {};
    long long fam_launches[F_COUNT] = {};
    std::vector<unsigned char> fam_of;   // family of bracket k (profiling pass only)
    // RS_PROF_SHAPES=1 (debugging aid): the profiling pass also keeps one "family M N K" tag per bracket and the engine prints
    // the time per distinct shape to stderr
    std::vector<std::string> tag_of;
    std::vector<double> tag_flops;
    // which network the launches belong to (rs_profile_shapes: "encoder" / "unet" / "decoder") and, profiling passes only, an event at every
    // change of part: the wall time of the parts between them (every kernel, not just the bracketed MFMA family)
    const char* part = "";
    std::vector<std::pair<std::string, hipEvent_t>> part_marks;
    void enter_part(const char* name) {
        part = name;
        if (dry || !prof || !prof->on) return;
        hipEvent_t ev = nullptr;
        if (hipEventCreate(&ev) != hipSuccess) return;
        (void)hipEventRecord(ev, st);
        part_marks.emplace_back(name, ev);
    }
    void fam_note(int f, double flops, long long M = 0, int N = 0, int K = 0, int nz = 1) {
        fam_flops[f] += flops; ++fam_launches[f];
        if (prof && prof->on) {
            fam_of.push_back((unsigned char)f);
            char b[128]; snprintf(b, sizeof b, "%s f%d M=%lld N=%d K=%d z=%d", part, f, M, N, K, nz);
            tag_of.emplace_back(b); tag_flops.push_back(flops);
        }
    }
    void igemm(const IGemmParams& p, int in_dt, int out_dt, int nz, const char* what) {
        const int Kall = p.Ktot + (p.sx ? p.sC : 0);   // (+ the K columns of a folded 1x1 shortcut)
        igemm_flops[in_dt == RS_F16 ? 0 : (in_dt == RS_F16S ? 2 : 1)] += 2.0 * (double)p.M * (double)p.Cout * (double)Kall * (double)nz;
        {
            int tw, bc;
            const bool wino = rs_wino_plan(&p, in_dt, out_dt, nz) != 0;
            const bool halo = !wino && rs_igemm4_pick(&p, in_dt, out_dt, nz, &tw, &bc) != 0;
            const int f = wino ? F_WINO_S : (in_dt == RS_F16 ? (halo ? F_HALO16 : F_IGEMM16) : (in_dt == RS_F16S ? (halo ? F_HALO_SPLIT : F_IGEMM_SPLIT) : F_IGEMM32));
            fam_note(f, 2.0 * (double)p.M * (double)p.Cout * (double)Kall * (double)nz, p.M, p.Cout, Kall, nz);
        }
        {
            const double isz = in_dt == RS_F16 ? 2.0 : 4.0, osz = out_dt == RS_F16 ? 2.0 : 4.0;
            const double src = (double)p.B * p.Hs * p.Ws * (double)(p.C0 + p.C1 + (p.sx ? p.sC : 0)) * isz;
            const double out = (double)p.M * p.Cout * osz;
            igemm_bytes += (double)nz * (src + (double)p.Cout * Kall * isz + out + (p.res ? out : 0.0));
        }
        ++igemm_launches;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof && prof->on) {
            if (prof->used + 2 > prof->ev.size()) {
                const size_t old = prof->ev.size();
                prof->ev.resize(old + 1024);
                for (size_t i = old; i < prof->ev.size(); ++i) (void)hipEventCreate(&prof->ev[i]);
            }
            e0 = prof->ev[prof->used++]; e1 = prof->ev[prof->used++];
            (void)hipEventRecord(e0, st);
        }
        check(rs_igemm_launch(&p, in_dt, out_dt, nz, st), what);
        if (p.splitk > 1) ++launches;   // (the slices' reduce kernel)
        if (e1) (void)hipEventRecord(e1, st);
    }
    // fused qkv projection + window attention: the projection's FLOPs / compulsory bytes stay in the MFMA-family bookkeeping
    void win_attn_qkv(const WinAttnParams& p, int E, int dt = RS_F16) {
        const double M = (double)p.B * p.H * p.W;
        const int sp = dt == RS_F16S;
        igemm_flops[sp ? 2 : 0] += 2.0 * M * (p.wproj ? 4.0 : 3.0) * E * E;
        fam_note(sp ? F_WINATTN_S : F_WINATTN, 2.0 * M * (p.wproj ? 4.0 : 3.0) * E * E + 2.0 * 2.0 * M * 64.0 * E, (long long)M, E, E);   // + QK^T and PV of the 64-token windows
        igemm_bytes += (sp ? 4.0 : 2.0) * (M * E * (p.res ? 3.0 : 2.0) + (p.wproj ? 4.0 : 3.0) * E * E);
        ++igemm_launches;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof && prof->on) {
            if (prof->used + 2 > prof->ev.size()) {
                const size_t old = prof->ev.size();
                prof->ev.resize(old + 1024);
                for (size_t i = old; i < prof->ev.size(); ++i) (void)hipEventCreate(&prof->ev[i]);
            }
            e0 = prof->ev[prof->used++]; e1 = prof->ev[prof->used++];
            (void)hipEventRecord(e0, st);
        }
        if (sp) check(rs_win_attn_qkv_split_launch(&p, st), "win_attn_qkv_split");
        else check(rs_win_attn_qkv_launch(&p, st), "win_attn_qkv");
        if (e1) (void)hipEventRecord(e1, st);
    }
    // streaming AE attention (ae_attn.hip): QK^T and PV of nz images, q / k / v^T read once, o written once
    void ae_flash(const void* q, int ldq, const void* k, int ldk, const void* vt, const float* bv, void* o, int ldo, int nz, int T, int C, float scale,
                  int dt = RS_F16) {
        const int sp = dt == RS_F16S;
        const double fl = 4.0 * (double)nz * (double)T * (double)T * (double)C;
        igemm_flops[sp ? 2 : 0] += fl;
        fam_note(sp ? F_AEFLASH_S : F_AEFLASH, fl, T, T, C, nz);
        igemm_bytes += (sp ? 4.0 : 2.0) * 4.0 * (double)nz * T * C;
        ++igemm_launches;
        hipEvent_t e0, e1;
        bracket(prof, st, e0, e1);
        if (sp) check(rs_ae_flash_split_launch(q, ldq, k, ldk, vt, bv, o, ldo, nz, T, C, scale, st), "ae_flash_attn_split");
        else check(rs_ae_flash_launch(q, ldq, k, ldk, vt, bv, o, ldo, nz, T, C, scale, st), "ae_flash_attn");
        if (e1) (void)hipEventRecord(e1, st);
    }
    // the fused Swin MLP belongs to the same MFMA family for the roofline bookkeeping: both GEMMs' FLOPs, compulsory bytes
    void swin_mlp(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y, int M, int ldx,
                  int ldres, int ldy, int E, int HD, const float* xcoef = nullptr, int HW = 0, int dt = RS_F16, float* ystats = nullptr,
                  int ystats_ld = 0, const GNTail* tail = nullptr, int NO = 0) {
        const int sp = dt == RS_F16S;
        if (NO <= 0) NO = E;   // (NO != E: patch_unembed folded in - fc1, then the product matrix over [h ; x])
        const double fl = 2.0 * (double)M * ((double)E * HD + (double)NO * (NO != E ? HD + E : HD));
        igemm_flops[sp ? 2 : 0] += fl;
        fam_note(sp ? F_SWINMLP_S : F_SWINMLP, fl, M, NO, HD);
        igemm_bytes += (sp ? 4.0 : 2.0) * ((double)M * (E * ((res || NO != E) ? 2.0 : 1.0) + NO) + (double)E * HD + (double)NO * (NO != E ? HD + E : HD));
        ++igemm_launches;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof && prof->on) {
            if (prof->used + 2 > prof->ev.size()) {
                const size_t old = prof->ev.size();
                prof->ev.resize(old + 1024);
                for (size_t i = old; i < prof->ev.size(); ++i) (void)hipEventCreate(&prof->ev[i]);
            }
            e0 = prof->ev[prof->used++]; e1 = prof->ev[prof->used++];
            (void)hipEventRecord(e0, st);
        }
        if (sp) check(rs_swin_mlp_split_launch_n(x, w1, b1, w2, b2, res, y, M, ldx, ldres, ldy, E, HD, NO, xcoef, HW, ystats, ystats_ld, tail, st), "swin_mlp_split");
        else check(rs_swin_mlp_launch(x, w1, b1, w2, b2, res, y, M, ldx, ldres, ldy, E, HD, xcoef, HW, ystats, ystats_ld, st), "swin_mlp");
        if (e1) (void)hipEventRecord(e1, st);
    }
};

}  // namespace

// Test hook, compiled ONLY into the test-hooks library (-DRS_TEST_HOOKS: resshift_amd.build.build_testhooks(), loaded by tests/_fake_device_plumbing.py
// alone - the production libresshift_hip.so does not contain it): with RS_FAKE_DEVICE=1 on a host WITHOUT a HIP device the engine's real pass walks
// its control flow on a host-memory arena while every launch fails - a CPU test compares its bookkeeping with the dry pass.
#ifdef RS_TEST_HOOKS
static bool rs_fake_device() {
    static const bool fake = []() {
        if (!getenv("RS_FAKE_DEVICE")) return false;
        int n = 0;
        const hipError_t e = hipGetDeviceCount(&n);
        // (only the definite answer "this host has no HIP device" enables the hook - a driver hiccup on a GPU box must not)
        return e == hipErrorNoDevice || (e == hipSuccess && n == 0);
    }();
    return fake;
}
#else
static constexpr bool rs_fake_device() { return false; }
#endif

struct rs_engine {
    rs_config cfg;
    std::unordered_map<std::string, HostTensor> host;
    Blob blob;
    size_t blob_bytes = 0;
    bool bound = false, ready = false;
    std::string build_err;
    // Split precision is optional per checkpoint: the halo kernel scales the hi weight fragment by 2^11 in fp16, which is exact only
    // for |w| < 32 (igemm4.hip).  A checkpoint with a larger (or non-finite) conv / linear weight still loads and runs in fp16 /
    // fp32; only a call that asks for RS_PREC_SPLIT fails.  The verdict travels IN the blob (first word of a 256-byte header), so
    // ranks that receive the blob by broadcast know it too.
    std::string split_err;
    bool split_ok = true;
    // ... and a layer whose weights reach |w| >= 30 is not a reason to refuse the policy: only the kernels that scale the hi fragment by
    // 2^11 (the halo conv, the fused split Swin kernels) cannot take it, the generic split kernel (igemm_split.hip: two accumulators, no
    // scaling) can.  One flag byte per conv / linear in build order, written into the blob by the packing rank (it travels with the
    // broadcast) and read back by rs_weights_ready; halo_conv() / basiclayer() route a flagged layer to the generic kernels.
    std::vector<unsigned char> big_w;
    int conv_count = 0;
    unsigned char* big_w_dev = nullptr;
    bool big(const ConvW& c) const { return c.idx >= 0 && c.idx < (int)big_w.size() && big_w[c.idx] != 0; }
    Arena arena;
    long long last_launches = 0;
    Exec::Prof prof, prof_gn;
    double last_flops[3] = {0.0, 0.0, 0.0}, last_igemm_ms = 0.0, last_igemm_bytes = 0.0, last_gn_ms = 0.0, last_gn_bytes = 0.0;
    double last_fam[Exec::F_COUNT][3] = {};   // flops, ms, launches per kernel family
    std::string last_shapes;                  // rs_profile_shapes: one line per (part, family, M, N, K) and per part of the last profiled call
    long long last_igemm_launches = 0, last_gn_launches = 0;
    bool debug = false;
    std::vector<std::pair<std::string, View>> trace;
    std::vector<TailPlan> tail_plan;   // made by the dry pass of a call, read by its real pass (Exec::plan)
    // UNet
    std::vector<UBlock> in_blocks, out_blocks;
    ResBlockW mid_res1, mid_res2; BasicLayerW mid_swin;
    std::vector<ConvW> fe_convs, fe_downs;
    GNW out_norm; ConvW out_conv;
    ConvW te0, te2;  // time_embed linears
    std::vector<int> skip_ch, h_ch;  // per input block / per output block
    int film_total = 0, fe_out_ch = 0;
    std::vector<ResBlockW*> film_blocks;
    std::map<int, float*> film_cache;
    // AE
    ConvW enc_in, enc_out, dec_in, dec_out, quant_conv, post_quant_conv;
    std::vector<AELevel> enc_levels, dec_levels;
    ResBlockW enc_mid1, enc_mid2, dec_mid1, dec_mid2;
    AttnW enc_attn, dec_attn;
    GNW enc_norm, dec_norm;
    float* codebook = nullptr;

    // ---------------------------------------------------------------- build
    // tensors the packer derives from checkpoint tensors (products of two linear maps that run as one GEMM): made on first use
    std::map<std::string, std::function<bool(HostTensor&)>> derived;
    const HostTensor* find(const std::string& k) {
        auto it = host.find(k);
        if (it == host.end()) {
            auto d = derived.find(k);
            if (d != derived.end()) {
                HostTensor t;
                if (d->second(t)) it = host.emplace(k, std::move(t)).first;
            }
        }
        if (it == host.end()) { if (build_err.empty()) build_err = "missing state_dict key: " + k; return nullptr; }
        return &it->second;
    }
    float* add_f32(const std::string& key, size_t n) {
        return (float*)blob.add(n * sizeof(float), [&](char* dst) {
            const HostTensor* t = find(key);
            if (!t) return;
            if (t->data.size() != n) { if (build_err.empty()) build_err = "bad size for " + key; return; }
            memcpy(dst, t->data.data(), n * sizeof(float));
        });
    }
    // Every conv runs on the MFMA implicit GEMM; input channels are zero-padded to a multiple of 8 (CinP) so that the
    // 16-byte K chunks stay aligned (3/6-channel image and latent inputs become 8-channel tensors).  `force_direct`
    // keeps the scalar kernel for the fp32-in/fp32-out 1x1 quant convs and for two-source convs whose first source is
    // not chunk aligned.
    // `head`: an output head (3x3, <= 4 output channels, Cin % 8 == 0): ALSO the fp32 [tap][Cin][Cout] form for the fused GroupNorm + SiLU +
    // conv kernel (direct_conv.hip: gn_silu_head_conv_kernel)
    ConvW add_conv(const std::string& prefix, int Cin, int Cout, int KH, int KW, bool has_bias = true, bool force_direct = false, bool head = false) {
        ConvW c; c.Cin = Cin; c.Cout = Cout; c.KH = KH; c.KW = KW;
        c.idx = conv_count++;
        if ((int)big_w.size() < conv_count) big_w.resize(conv_count, 0);
        const int cidx = c.idx;
        c.direct = force_direct;
        c.CinP = c.direct ? Cin : (Cin + 7) / 8 * 8;
        const int CinP = c.CinP;
        const size_t K = (size_t)KH * KW * CinP, n = (size_t)KH * KW * Cin * Cout, np = K * Cout;
        const std::string wkey = prefix + ".weight";
        auto get = [&]() -> const float* {
            const HostTensor* t = find(wkey);
            if (!t) return nullptr;
            if (t->data.size() != n) { if (build_err.empty()) build_err = "bad size for " + wkey; return nullptr; }
            return t->data.data();
        };
        if (c.direct) {
            c.wd = (float*)blob.add(n * 4, [&](char* dst) {
                const float* w = get(); if (!w) return;
                float* o = (float*)dst;  // [K][Cout], k = (ky*KW+kx)*Cin + ci
                for (int co = 0; co < Cout; ++co)
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int t = 0; t < KH * KW; ++t)
                            o[((size_t)t * Cin + ci) * Cout + co] = w[((size_t)co * Cin + ci) * KH * KW + t];
            });
        } else {
            // (the staging buffer is zero-initialised, so padded input channels keep zero weights)
            if (cfg.enable_f16)
                c.wh = blob.add(np * 2, [&](char* dst) {
                    const float* w = get(); if (!w) return;
                    f16* o = (f16*)dst;  // [Cout][K]
                    for (int co = 0; co < Cout; ++co)
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int t = 0; t < KH * KW; ++t) {
                                const float wv = w[((size_t)co * Cin + ci) * KH * KW + t];
                                // (a derived weight - the sub-pixel form's summed taps - can reach 4 |w|: outside the fp16 range it would become inf silently)
                                if (!(std::fabs(wv) <= 65504.0f) && build_err.empty()) build_err = "fp16 storage needs weights inside the fp16 range: " + wkey;
                                o[(size_t)co * K + (size_t)t * CinP + ci] = (f16)wv;
                            }
                });
            if (cfg.enable_split)
                c.ws = blob.add(np * 4, [&](char* dst) {
                    const float* w = get(); if (!w) return;
                    f16* o = (f16*)dst;  // [Cout][K hi | K lo], lo = (w - hi) * 2^11 (common.h: split storage)
                    for (int co = 0; co < Cout; ++co)
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int t = 0; t < KH * KW; ++t) {
                                f16 h, l;
                                const float wv = w[((size_t)co * Cin + ci) * KH * KW + t];
                                // (the halo kernel and the fused Swin kernels scale the hi fragment by 2^11 in fp16: exact below 32 - a layer
                                // beyond that runs on the generic split kernel; a non-finite weight rules the policy out)
                                if (!std::isfinite(wv) || std::fabs(wv) > 60000.0f) { if (split_err.empty()) split_err = "split precision needs finite fp16-range weights: " + wkey; }
                                else if (!(std::fabs(wv) < 30.0f)) big_w[cidx] = 1;
                                rs_split(wv, h, l);
                                const size_t k = (size_t)t * CinP + ci;
                                o[(size_t)co * 2 * K + k] = h;
                                o[(size_t)co * 2 * K + K + k] = l;
                            }
                });
            // Winograd F(2x2,3x3) form (wino.hip).  RS_WINO=0 when the engine is CREATED switches it off (the blob layout depends on it: every rank
            // of a run has to agree, like RS_UPFOLD).  Only the layers whose whole output fits 160-channel blocks (the UNet's 160 / 320-channel ResBlock
            // convs, + 200 MB of blob): on those the kernel measured 1.07 - 1.12 x the halo kernel - 243.6 -> 238.6 ms per parity pass on one box,
            // two pairs - on the autoencoder's 128 / 256 / 512-channel layers 0.92 - 1.04 x (profiles/r6_wino_bench.txt).
            static const bool wino_on = []() { const char* e = getenv("RS_WINO"); return !(e && e[0] == '0'); }();
            if (wino_on && cfg.enable_split && KH == 3 && KW == 3 && (Cin % 32) == 0 && Cin <= 640 && (Cout % 160) == 0 && Cout <= 320)
                c.ww = blob.add(rs_wino_weight_bytes(Cin, Cout), [&](char* dst) {
                    const float* w = get(); if (!w) return;
                    if (!(rs_wino_pack(w, Cin, Cout, dst) < 30.0f)) big_w[cidx] = 1;   // (the kernel scales the hi fragment by 2^11 in fp16, like the halo kernel)
                });
            if (cfg.enable_f32)
                c.wf = blob.add(np * 4, [&](char* dst) {
                    const float* w = get(); if (!w) return;
                    float* o = (float*)dst;
                    for (int co = 0; co < Cout; ++co)
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int t = 0; t < KH * KW; ++t)
                                o[(size_t)co * K + (size_t)t * CinP + ci] = w[((size_t)co * Cin + ci) * KH * KW + t];
                });
        }
        if (head && !c.direct && KH == 3 && KW == 3 && Cout <= 4 && (Cin % 8) == 0)
            c.wd = (float*)blob.add(n * 4, [&](char* dst) {
                const float* w = get(); if (!w) return;
                float* o = (float*)dst;  // [tap][Cin][Cout]
                for (int co = 0; co < Cout; ++co)
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int t = 0; t < 9; ++t) o[((size_t)t * Cin + ci) * Cout + co] = w[((size_t)co * Cin + ci) * 9 + t];
            });
        if (has_bias) c.bias = add_f32(prefix + ".bias", Cout);
        return c;
    }
    // Sub-pixel form of "nearest x2 upsample, then conv3x3" (models/unet.py:53-81 Upsample, ldm/modules/diffusionmodules/model.py:50-65):
    // every output pixel (2y + py, 2x + px) sees only a 2 x 2 neighbourhood of the LOW-resolution input, because the taps that land on the
    // same source pixel can be added up front - rows: py = 0: {y - 1: w[0], y: w[1] + w[2]}, py = 1: {y: w[0] + w[1], y + 1: w[2]}, columns
    // alike; zero padding of the upsampled image IS zero padding of the source (rows -1 and 2H map to -1 and H).  Four 2x2 convs (one per
    // output parity, pad_t = 1 - py, pad_l = 1 - px) with K = 4 Cin instead of one 3x3 conv with K = 9 Cin on four times the pixels: 2.25 x
    // fewer multiply-adds, the same result up to the rounding of the summed weights (formed in double from the checkpoint's tensor, like the
    // other derived matrices).  The generic kernels scatter their rows into the big tensor (IGemmParams::osc).  RS_UPFOLD=0: off.
    bool add_upfold(const std::string& prefix, int C, ConvW (&upf)[4]) {
        static const bool on = []() { const char* e = getenv("RS_UPFOLD"); return !(e && e[0] == '0'); }();
        if (!on || (C % 8)) return false;
        for (int q = 0; q < 4; ++q) {
            const int py = q >> 1, px = q & 1;
            const std::string fk = prefix + ".upfold" + std::to_string(q);
            derived[fk + ".weight"] = [this, prefix, C, py, px](HostTensor& t) {
                const HostTensor* w = find(prefix + ".weight");
                if (!w || w->data.size() != (size_t)C * C * 9) return false;
                t.data.assign((size_t)C * C * 4, 0.f);
                t.shape = {C, C, 2, 2};
                // source offset dy in {0, 1} of parity py collects the taps ky with ((py + ky - 1) >> 1) - (py ? 0 : -1) == dy
                auto taps = [](int par, int d, int (&k)[2]) -> int {   // taps of one axis that land on source offset d (relative to y - 1 + par)
                    int n = 0;
                    for (int kk = 0; kk < 3; ++kk) {
                        const int src = (par + kk - 1) >> 1;            // relative to y (arithmetic shift: -1 >> 1 = -1)
                        if (src - (par - 1) == d) k[n++] = kk;
                    }
                    return n;
                };
                for (int co = 0; co < C; ++co)
                    for (int ci = 0; ci < C; ++ci) {
                        const float* w9 = w->data.data() + ((size_t)co * C + ci) * 9;
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx) {
                                int ky[2], kx[2];
                                const int ny = taps(py, dy, ky), nx = taps(px, dx, kx);
                                double a = 0.0;
                                for (int i = 0; i < ny; ++i)
                                    for (int j = 0; j < nx; ++j) a += (double)w9[ky[i] * 3 + kx[j]];
                                t.data[((size_t)co * C + ci) * 4 + dy * 2 + dx] = (float)a;
                            }
                    }
                return true;
            };
            derived[fk + ".bias"] = [this, prefix, C](HostTensor& t) {
                const HostTensor* b = find(prefix + ".bias");
                if (!b || b->data.size() != (size_t)C) return false;
                t = *b;
                return true;
            };
            upf[q] = add_conv(fk, C, C, 2, 2);
        }
        return true;
    }
    // the four launches of the sub-pixel form: x [B,H,W,C] -> y [B,2H,2W,C].  Taken when the low-resolution grid alone fills the chip
    // (the 8 x 8 -> 16 x 16 and 16 x 16 -> 32 x 32 steps at batch 32 do not: they keep the folded-address form).
    bool upfold_ok(const Exec& ex, const ConvW (&upf)[4], const View& x, const View& y) const {
        if (ex.dbg || x.dt != y.dt || (x.dt != RS_F16 && x.dt != RS_F16S) || y.H != 2 * x.H || y.W != 2 * x.W || x.C != upf[0].CinP) return false;
        if (!upf[0].w_for(x.dt) || (x.ld % 8) || (y.ld % 8)) return false;   // (the scattered-row launches' alignment preconditions)
        // (measured, profiles/r5_upfold_ab.txt: with the 16 -> 32 and 8 -> 16 steps as well - 8192 / 2048 low-resolution pixels at batch 32, four
        // launches that cannot fill the chip each - the pass is 0.9 ms slower than with the 32 -> 64 step alone)
        return (long long)x.B * x.H * x.W >= 16384;
    }
    // statistics of y for the consuming GroupNorm from the four launches' epilogues (split storage): one slab per low-resolution pixel tile
    // and parity class
    void upfold_want_stats(Exec& ex, const ConvW (&upf)[4], const View& x, View& y) {
        static const bool on = []() { const char* e = getenv("RS_GN_EPI_STATS"); return !(e && e[0] == '0'); }();
        static const bool gen = []() { const char* e = getenv("RS_GN_GEN_STATS"); return !(e && e[0] == '0'); }();
        y.st = nullptr; y.st2 = nullptr; y.st_prod = -1;
        if (!on || !gen || ex.dbg || x.dt != RS_F16S || y.dt != RS_F16S) return;
        View yl = y; yl.H = x.H; yl.W = x.W;
        const IGemmParams pp = conv_params(upf[0], x, nullptr, yl, 1, 1, 1, 1, 0, nullptr, 1.f);
        const int spx = rs_igemm_split_stats_px(&pp, 1);
        if (spx <= 0 || ((x.H * x.W) % spx)) return;
        y.stS = 4 * (x.H * x.W / spx); y.stld = y.C;
        y.st = ex.pool((size_t)y.B * y.stS * y.stld * 2 * sizeof(float));
        y.st_prod = ex.prod_seq++;
    }
    void upfold_conv(Exec& ex, const ConvW (&upf)[4], const View& x, const View& y) {
        if (ex.dry) return;
        GNTail tl{};
        if (y.st) (void)ex.fill_tail(y.st_prod, y.B, tl);   // ONE tail (one ticket per image) for the four launches
        for (int q = 0; q < 4; ++q) {
            const int py = q >> 1, px = q & 1;
            View yl = y; yl.H = x.H; yl.W = x.W;   // the launch computes the low-resolution grid; its rows are scattered into y
            IGemmParams p = conv_params(upf[q], x, nullptr, yl, 1, 1 - py, 1 - px, 1, 0, nullptr, 1.f);
            p.no_halo = 1;
            p.osc = 2; p.ooy = py; p.oox = px;
            if (y.st) { p.ystats = y.st; p.ystats_ld = y.stld; p.tail = tl; }
            if (!p.w) { ex.err = -3; g_err = "weights for this precision were not packed (enable_f16/enable_f32/enable_split)"; return; }
            ex.igemm(p, x.dt, y.dt, 1, "igemm");
        }
    }
    // Fragment-major copies of a 1x1 weight [N][K] (N % 16 == 0, K % 32 == 0) for win_attn_qkv_kernel / win_attn_qkv_split_kernel: lane
    // (lr, lg) of the wave that multiplies rows 16 nb .. 16 nb + 15 with k step ks reads W[16 nb + lr][32 ks + 8 lg .. + 7] - from the
    // row-major weight that is 16 different cache lines per wave instruction, from this copy 8 full ones.
    void add_frag_copies(ConvW& c, const std::string& prefix) {
        const int N = c.Cout, K = c.Cin;
        if (c.KH != 1 || c.KW != 1 || (N % 16) || (K % 32)) return;
        const std::string wkey = prefix + ".weight";
        const size_t n = (size_t)N * K;
        auto get = [this, wkey, n]() -> const float* {
            const HostTensor* t = find(wkey);
            return (t && t->data.size() == n) ? t->data.data() : nullptr;
        };
        if (cfg.enable_f16)
            c.wh_frag = blob.add(n * 2, [=](char* dst) {
                const float* w = get(); if (!w) return;
                rs_pack_frag_major(w, N, K, (f16*)dst, nullptr);
            });
        if (cfg.enable_split)
            c.ws_frag = blob.add(n * 4, [=](char* dst) {
                const float* w = get(); if (!w) return;
                rs_pack_frag_major(w, N, K, nullptr, (f16*)dst);
            });
    }
    // plain fp32 linear kept in the reference [N][K] layout (time embedding MLP, emb_layers)
    ConvW add_linear_f32(const std::string& prefix, int K, int N) {
        ConvW c; c.Cin = K; c.Cout = N;
        c.wd = add_f32(prefix + ".weight", (size_t)K * N);
        c.bias = add_f32(prefix + ".bias", N);
        return c;
    }
    GNW add_gn(const std::string& prefix, int C) {
        GNW g; g.C = C; g.gamma = add_f32(prefix + ".weight", C); g.beta = add_f32(prefix + ".bias", C); return g;
    }
    ResBlockW add_resblock(const std::string& p, int Cin, int Cout, int emb_ch) {
        ResBlockW r; r.Cin = Cin; r.Cout = Cout;
        r.n1 = add_gn(p + ".in_layers.0", Cin);
        r.c1 = add_conv(p + ".in_layers.2", Cin, Cout, 3, 3);
        r.emb = add_linear_f32(p + ".emb_layers.1", emb_ch, 2 * Cout);
        r.n2 = add_gn(p + ".out_layers.0", Cout);
        r.c2 = add_conv(p + ".out_layers.3", Cout, Cout, 3, 3);
        r.has_skip = Cin != Cout;
        if (r.has_skip) r.skip = add_conv(p + ".skip_connection", Cin, Cout, 1, 1);
        r.film_off = film_total; film_total += 2 * Cout;
        return r;
    }
    BasicLayerW add_basiclayer(const std::string& p, int C, int ds) {
        const rs_unet_config& u = cfg.unet;
        BasicLayerW b; b.C = C; b.E = u.swin_embed_dim;
        const int E = b.E, heads = u.num_heads, hidden = (int)(E * u.mlp_ratio);
        b.embed = add_conv(p + ".patch_embed.proj", C, E, 1, 1);
        for (int d = 0; d < u.swin_depth; ++d) {
            const std::string q = p + ".blocks." + std::to_string(d);
            SwinBlockW s;
            // shift_size is fixed at construction from the *constructed* resolution (swin_transformer.py:189-194)
            s.shift = (d % 2 == 1 && ds > u.window_size) ? u.window_size / 2 : 0;
            s.n1 = add_gn(q + ".norm1", E);
            s.qkv = add_conv(q + ".attn.qkv", E, 3 * E, 1, 1);
            add_frag_copies(s.qkv, q + ".attn.qkv");
            const std::string tkey = q + ".attn.relative_position_bias_table";
            s.bias_t = (float*)blob.add((size_t)heads * 64 * 64 * 4, [&, tkey, heads](char* dst) {
                const HostTensor* t = find(tkey);
                if (!t) return;
                if ((int)t->data.size() != 225 * heads) { if (build_err.empty()) build_err = "bad size for " + tkey; return; }
                float* o = (float*)dst;  // [h][j][i]; index of (i,j): swin_transformer.py:93-102
                for (int h = 0; h < heads; ++h)
                    for (int j = 0; j < 64; ++j)
                        for (int i = 0; i < 64; ++i) {
                            const int idx = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
                            o[((size_t)h * 64 + j) * 64 + i] = t->data[(size_t)idx * heads + h];
                        }
            });
            s.bias_n = (float*)blob.add((size_t)heads * 64 * 64 * 4, [&, tkey, heads](char* dst) {
                const HostTensor* t = find(tkey);
                if (!t || (int)t->data.size() != 225 * heads) return;
                float* o = (float*)dst;  // [h][i][j] for the MFMA window kernel
                for (int h = 0; h < heads; ++h)
                    for (int i = 0; i < 64; ++i)
                        for (int j = 0; j < 64; ++j) {
                            const int idx = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
                            o[((size_t)h * 64 + i) * 64 + j] = t->data[(size_t)idx * heads + h];
                        }
            });
            // compact form for the split fused kernel (WinAttnParams::bias_c): the table itself, head-major, in units of log2
            s.bias_c = (float*)blob.add((size_t)heads * 256 * 4, [&, tkey, heads](char* dst) {
                const HostTensor* t = find(tkey);
                float* o = (float*)dst;
                std::fill(o, o + (size_t)heads * 256, 0.0f);
                if (!t || (int)t->data.size() != 225 * heads) return;
                for (int h = 0; h < heads; ++h)
                    for (int k = 0; k < 225; ++k) o[h * 256 + k] = t->data[(size_t)k * heads + h] * 1.44269504088896f;
            });
            s.proj = add_conv(q + ".attn.proj", E, E, 1, 1);
            add_frag_copies(s.proj, q + ".attn.proj");
            s.n2 = add_gn(q + ".norm2", E);
            s.fc1 = add_conv(q + ".mlp.fc1", E, hidden, 1, 1);
            s.fc2 = add_conv(q + ".mlp.fc2", hidden, E, 1, 1);
            b.blocks.push_back(s);
        }
        b.unembed = add_conv(p + ".patch_unembed.proj", E, C, 1, 1);
        // patch_unembed folded into the last block's fused split MLP (swin_mlp.hip, NO != E): y = Wu (x + W2 h + b2) + bu
        //   = [Wu W2 | Wu] [h ; x] + (Wu b2 + bu) - a [C][hidden + E] matrix and a C-vector, products in double from the checkpoint's tensors
        if (cfg.enable_split && u.swin_depth > 0 && rs_swin_mlp_split_unembed_supported(E, hidden, C)) {
            const std::string fk = p + ".patch_unembed.fold", uk = p + ".patch_unembed.proj", mk = p + ".blocks." + std::to_string(u.swin_depth - 1) + ".mlp.fc2";
            derived[fk + ".weight"] = [this, uk, mk, E, hidden, C](HostTensor& t) {
                const HostTensor* wu = find(uk + ".weight"); const HostTensor* w2 = find(mk + ".weight");
                if (!wu || !w2 || wu->data.size() != (size_t)C * E || w2->data.size() != (size_t)E * hidden) return false;
                t.data.assign((size_t)C * (hidden + E), 0.f);
                t.shape = {C, hidden + E, 1, 1};
                std::vector<double> row(hidden);
                for (int n = 0; n < C; ++n) {
                    std::fill(row.begin(), row.end(), 0.0);
                    for (int e = 0; e < E; ++e) {
                        const double a = wu->data[(size_t)n * E + e];
                        const float* w2r = w2->data.data() + (size_t)e * hidden;
                        for (int h = 0; h < hidden; ++h) row[h] += a * (double)w2r[h];
                    }
                    float* o = t.data.data() + (size_t)n * (hidden + E);
                    for (int h = 0; h < hidden; ++h) o[h] = (float)row[h];
                    for (int e = 0; e < E; ++e) o[hidden + e] = wu->data[(size_t)n * E + e];
                }
                return true;
            };
            derived[fk + ".bias"] = [this, uk, mk, E, C](HostTensor& t) {
                const HostTensor* wu = find(uk + ".weight"); const HostTensor* bu = find(uk + ".bias"); const HostTensor* b2 = find(mk + ".bias");
                if (!wu || !bu || !b2 || wu->data.size() != (size_t)C * E || bu->data.size() != (size_t)C || b2->data.size() != (size_t)E) return false;
                t.data.resize(C); t.shape = {C};
                for (int n = 0; n < C; ++n) {
                    double a = bu->data[n];
                    for (int e = 0; e < E; ++e) a += (double)wu->data[(size_t)n * E + e] * (double)b2->data[e];
                    t.data[n] = (float)a;
                }
                return true;
            };
            b.unfold = add_conv(fk, hidden + E, C, 1, 1);
            b.has_unfold = true;
        }
        return b;
    }
    bool in_attn_res(int ds) const {
        for (int i = 0; i < cfg.unet.n_attn_res; ++i) if (cfg.unet.attention_resolutions[i] == ds) return true;
        return false;
    }
    void build_unet() {
        const rs_unet_config& u = cfg.unet;
        in_blocks.clear(); out_blocks.clear(); fe_convs.clear(); fe_downs.clear(); skip_ch.clear(); h_ch.clear();
        film_total = 0;
        const int mc = u.model_channels, emb_ch = 4 * mc;
        te0 = add_linear_f32("time_embed.0", mc, emb_ch);
        te2 = add_linear_f32("time_embed.2", emb_ch, emb_ch);
        int base_chn;
        if (u.cond_lq && u.lq_size == u.image_size) {
            base_chn = u.cond_mask ? 4 : 3;
        } else {
            int feature_chn = u.cond_mask ? 4 : 3;
            base_chn = 16;
            const int stages = (int)std::lround(std::log2((double)u.lq_size / u.image_size));
            for (int ii = 0; ii < stages; ++ii) {
                fe_convs.push_back(add_conv("feature_extractor." + std::to_string(3 * ii), feature_chn, base_chn, 3, 3));
                fe_downs.push_back(add_conv("feature_extractor." + std::to_string(3 * ii + 2) + ".op", base_chn, base_chn * 2, 3, 3));
                base_chn *= 2;
                feature_chn = base_chn;
            }
        }
        fe_out_ch = u.cond_lq ? base_chn : 0;
        int ch = u.channel_mult[0] * mc;
        const int input_ch = ch;
        {
            UBlock b; b.has_conv = true; b.level = 0; b.out_ch = ch;
            // with a feature extractor the conv reads two sources (x | features): both must be 16-byte chunk aligned
            const bool two_src_unaligned = !fe_convs.empty() && (u.in_channels % 8 != 0);
            b.conv = add_conv("input_blocks.0.0", u.in_channels + fe_out_ch, ch, 3, 3, true, two_src_unaligned);
            in_blocks.push_back(b);
        }
        std::vector<int> chans{ch};
        int ds = u.image_size;
        for (int level = 0; level < u.n_levels; ++level) {
            const int mult = u.channel_mult[level];
            for (int jj = 0; jj < u.num_res_blocks[level]; ++jj) {
                UBlock b; b.level = level;
                const std::string p = "input_blocks." + std::to_string(in_blocks.size());
                b.has_res = true; b.res = add_resblock(p + ".0", ch, mult * mc, emb_ch);
                ch = mult * mc;
                if (in_attn_res(ds) && jj == 0) { b.has_swin = true; b.swin = add_basiclayer(p + ".1", ch, ds); }
                b.out_ch = ch;
                in_blocks.push_back(b); chans.push_back(ch);
            }
            if (level != u.n_levels - 1) {
                UBlock b; b.level = level + 1; b.has_down = true; b.out_ch = ch;
                b.conv = add_conv("input_blocks." + std::to_string(in_blocks.size()) + ".0.op", ch, ch, 3, 3);
                in_blocks.push_back(b); chans.push_back(ch);
                ds /= 2;
            }
        }
        skip_ch = chans;
        mid_res1 = add_resblock("middle_block.0", ch, ch, emb_ch);
        mid_swin = add_basiclayer("middle_block.1", ch, ds);
        mid_res2 = add_resblock("middle_block.2", ch, ch, emb_ch);
        for (int level = u.n_levels - 1; level >= 0; --level) {
            const int mult = u.channel_mult[level];
            for (int i = 0; i <= u.num_res_blocks[level]; ++i) {
                const int ich = chans.back(); chans.pop_back();
                UBlock b; b.level = level;
                const std::string p = "output_blocks." + std::to_string(out_blocks.size());
                h_ch.push_back(ch);
                int sub = 0;
                b.has_res = true; b.res = add_resblock(p + "." + std::to_string(sub++), ch + ich, mc * mult, emb_ch);
                ch = mc * mult;
                if (in_attn_res(ds) && i == 0) { b.has_swin = true; b.swin = add_basiclayer(p + "." + std::to_string(sub++), ch, ds); }
                if (level && i == u.num_res_blocks[level]) {
                    b.has_up = true;
                    b.conv = add_conv(p + "." + std::to_string(sub) + ".conv", ch, ch, 3, 3);
                    b.has_upf = add_upfold(p + "." + std::to_string(sub++) + ".conv", ch, b.upf);
                    ds *= 2;
                }
                b.out_ch = ch;
                out_blocks.push_back(b);
            }
        }
        out_norm = add_gn("out.0", ch);
        out_conv = add_conv("out.2", input_ch, u.out_channels, 3, 3, true, false, /*head=*/true);
    }
    ResBlockW add_resnet(const std::string& p, int Cin, int Cout) {
        ResBlockW r; r.Cin = Cin; r.Cout = Cout;
        r.n1 = add_gn(p + ".norm1", Cin);
        r.c1 = add_conv(p + ".conv1", Cin, Cout, 3, 3);
        r.n2 = add_gn(p + ".norm2", Cout);
        r.c2 = add_conv(p + ".conv2", Cout, Cout, 3, 3);
        r.has_skip = Cin != Cout;
        if (r.has_skip) r.skip = add_conv(p + ".nin_shortcut", Cin, Cout, 1, 1);
        return r;
    }
    AttnW add_attn(const std::string& p, int C) {
        AttnW a; a.C = C;
        a.norm = add_gn(p + ".norm", C);
        a.q = add_conv(p + ".q", C, C, 1, 1);
        a.k = add_conv(p + ".k", C, C, 1, 1);
        a.v = add_conv(p + ".v", C, C, 1, 1);
        a.proj = add_conv(p + ".proj_out", C, C, 1, 1);
        return a;
    }
    void build_ae() {
        const rs_ae_config& a = cfg.ae;
        enc_levels.clear(); dec_levels.clear();
        // Encoder (model.py:452-547)
        enc_in = add_conv("encoder.conv_in", a.in_channels, a.ch, 3, 3);
        int block_in = a.ch;
        for (int l = 0; l < a.n_levels; ++l) {
            AELevel L;
            block_in = a.ch * (l == 0 ? 1 : a.ch_mult[l - 1]);
            const int block_out = a.ch * a.ch_mult[l];
            for (int i = 0; i < a.num_res_blocks[l]; ++i) {
                L.blocks.push_back(add_resnet("encoder.down." + std::to_string(l) + ".block." + std::to_string(i), block_in, block_out));
                block_in = block_out;
            }
            if (l != a.n_levels - 1) {
                L.has_resample = true;
                L.resample = add_conv("encoder.down." + std::to_string(l) + ".downsample.conv", block_in, block_in, 3, 3);
            }
            enc_levels.push_back(L);
        }
        enc_mid1 = add_resnet("encoder.mid.block_1", block_in, block_in);
        enc_attn = add_attn("encoder.mid.attn_1", block_in);
        enc_mid2 = add_resnet("encoder.mid.block_2", block_in, block_in);
        enc_norm = add_gn("encoder.norm_out", block_in);
        enc_out = add_conv("encoder.conv_out", block_in, a.z_channels, 3, 3, true, false, /*head=*/true);
        quant_conv = add_conv("quant_conv", a.z_channels, a.embed_dim, 1, 1, true, /*force_direct=*/true);  // fp32 in / fp32 out
        // Decoder (model.py:550-660)
        post_quant_conv = add_conv("post_quant_conv", a.embed_dim, a.z_channels, 1, 1, true, /*force_direct=*/true);  // fp32 VQ output in
        block_in = a.ch * a.ch_mult[a.n_levels - 1];
        dec_in = add_conv("decoder.conv_in", a.z_channels, block_in, 3, 3);
        dec_mid1 = add_resnet("decoder.mid.block_1", block_in, block_in);
        dec_attn = add_attn("decoder.mid.attn_1", block_in);
        dec_mid2 = add_resnet("decoder.mid.block_2", block_in, block_in);
        dec_levels.resize(a.n_levels);
        for (int l = a.n_levels - 1; l >= 0; --l) {
            AELevel L;
            const int block_out = a.ch * a.ch_mult[l];
            for (int i = 0; i <= a.num_res_blocks[l]; ++i) {
                L.blocks.push_back(add_resnet("decoder.up." + std::to_string(l) + ".block." + std::to_string(i), block_in, block_out));
                block_in = block_out;
            }
            if (l != 0) {
                L.has_resample = true;
                L.resample = add_conv("decoder.up." + std::to_string(l) + ".upsample.conv", block_in, block_in, 3, 3);
                L.has_upf = add_upfold("decoder.up." + std::to_string(l) + ".upsample.conv", block_in, L.upf);
            }
            dec_levels[l] = L;
        }
        dec_norm = add_gn("decoder.norm_out", block_in);
        dec_out = add_conv("decoder.conv_out", block_in, a.out_ch, 3, 3, true, false, /*head=*/true);
        codebook = add_f32("quantize.embedding.weight", (size_t)a.n_embed * a.embed_dim);
    }
    size_t build(char* base, bool fill) {
        blob.base = base; blob.off = 0; blob.fill = fill;
        build_err.clear();
        split_err.clear();
        if (fill) blob.staging.assign(blob_bytes, 0);
        (void)blob.add(256, [](char*) {});   // header: word 0 = flags (bit 0: split weights usable), written by rs_pack_weights
        conv_count = 0;
        if (fill) std::fill(big_w.begin(), big_w.end(), 0);
        if (cfg.has_unet) build_unet();
        if (cfg.has_ae) build_ae();
        // (the fillers run in order: by the time this one copies the table every split-weight filler has set its flag)
        big_w_dev = (unsigned char*)blob.add((size_t)conv_count, [&](char* dst) { memcpy(dst, big_w.data(), (size_t)conv_count); });
        return (blob.off + 255) & ~(size_t)255;
    }

    // ---------------------------------------------------------------- ops
    // parameter block of one implicit-GEMM conv launch (shared by conv() and the halo-kernel eligibility test)
    static IGemmParams conv_params(const ConvW& w, const View& x, const View* x1, const View& y, int stride, int pad_t, int pad_l, int up,
                                   int act, const View* res, float out_scale) {
        const int C1 = x1 ? x1->C : 0;
        IGemmParams p{};
        p.x0 = x.p; p.x1 = x1 ? x1->p : nullptr; p.w = w.w_for(x.dt); p.bias = w.bias;
        p.res = res ? res->p : nullptr; p.y = y.p;
        p.C0 = x.C; p.C1 = C1; p.ld0 = x.ld; p.ld1 = x1 ? x1->ld : 0;
        p.B = x.B; p.Hs = x.H; p.Ws = x.W; p.up = up; p.Ho = y.H; p.Wo = y.W; p.KH = w.KH; p.KW = w.KW;
        p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.Cout = w.Cout; p.ldy = y.ld; p.ldres = res ? res->ld : 0;
        p.M = y.B * y.H * y.W; p.Ktot = w.KH * w.KW * (x.C + C1); p.act = act; p.out_scale = out_scale;
        p.splitk = 1;
        p.ww = (x.dt == RS_F16S && !x1) ? w.ww : nullptr;
        return p;
    }
    // true when this 3x3 conv runs on the halo kernel (igemm4.hip), which can apply a GroupNorm affine + SiLU to its input
    // while the halo tile sits in LDS: the producer's raw output is read, the GroupNorm apply pass disappears
    // (`sk`: the halo kernel's own split-K factor for this launch - the small planes of the 16 x 16 / 8 x 8 levels run as split-K
    // slices over the stage sequence, igemm4_kernel.h; `seg`: its tile geometry, 8 = four 8 x 8 images per tile)
    // (`folded`: the launch will carry a folded 1x1 shortcut - only the halo kernel does that; `wino`: out, the Winograd kernel takes it)
    bool halo_conv(const ConvW& w, const View& x, const View& y, const View* res, int* sk = nullptr, int* seg = nullptr, bool folded = false,
                   bool* wino = nullptr) const {
        if (wino) *wino = false;
        if (w.direct || (x.dt != RS_F16 && x.dt != RS_F16S) || y.dt != x.dt || x.C != w.CinP) return false;
        if (x.dt == RS_F16S && big(w)) return false;   // |w| >= 30: no 2^11 scaling of the hi fragment - the generic split kernel takes it (conv(): IGemmParams::no_halo)
        const IGemmParams p = conv_params(w, x, nullptr, y, 1, 1, 1, 1, 0, res, 1.f);
        if (!folded && rs_wino_plan(&p, x.dt, y.dt, 1)) {   // same fusions as the halo kernel (input transform, statistics, tail), its own tiles
            if (sk) *sk = 1;
            if (seg) *seg = 0;
            if (wino) *wino = true;
            return true;
        }
        int tw, bc, sg = 0, k = 1;
        if (!rs_igemm4_plan(&p, x.dt, y.dt, 1, &tw, &bc, &sg, &k)) return false;
        if (sk) *sk = k;
        if (seg) *seg = sg;
        return true;
    }
    void conv(Exec& ex, const ConvW& w, const View& x, const View* x1, const View& y, int stride, int pad_t, int pad_l, int up,
              int act, const View* res, float out_scale = 1.f, const float* xcoef = nullptr, int xact = RS_ACT_NONE,
              const ConvW* skw = nullptr, const View* skx = nullptr) {
        const int C1 = x1 ? x1->C : 0;
        // split-K for launches that cannot fill the chip (8x8 / 16x16 UNet levels): fp32 slabs live in the arena
        int splitk = 1;
        float* partial = nullptr;
        if (!w.direct) {
            const int M = y.B * y.H * y.W;
            int sk4 = 1;
            // the halo kernel plans its own split-K (slices of the stage sequence); everything else asks the generic planner
            if (!x1 && w.KH == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && up == 1 && halo_conv(w, x, y, res, &sk4, nullptr, skw != nullptr)) splitk = sk4;
            else splitk = rs_igemm_splitk_plan(M, w.Cout, w.KH * w.KW * (x.C + C1), x.dt);
            if (splitk > 1) partial = (float*)ex.raw((size_t)splitk * M * w.Cout * sizeof(float));
        }
        // split storage has no two-source implicit GEMM: gather the channel concatenation once (only the first conv of the
        // feature-extractor configs, unet.py:882)
        View xcat;
        if (!w.direct && x1 && x.dt == RS_F16S) {
            xcat = ex.T(x.B, x.H, x.W, x.C + C1, x.dt);
            if (!ex.dry) {
                ex.check(rs_copy_channels_launch(x.p, x.ld, xcat.p, xcat.ld, x.C, x.pixels(), x.dt, ex.st), "concat copy");
                ex.check(rs_copy_channels_launch(x1->p, x1->ld, xcat.slice(x.C, C1).p, xcat.ld, C1, x.pixels(), x.dt, ex.st), "concat copy");
            }
            conv(ex, w, xcat, nullptr, y, stride, pad_t, pad_l, up, act, res, out_scale);
            return;
        }
        if (ex.dry) return;
        if (x.C + C1 != w.CinP) { if (!ex.err) { ex.err = -3; g_err = "conv input channels do not match the packed weights"; } return; }
        if (w.direct) {
            DirectConvParams p{};
            p.x0 = x.p; p.x1 = x1 ? x1->p : nullptr; p.w = w.wd; p.bias = w.bias; p.y = y.p;
            p.C0 = x.C; p.C1 = C1; p.ld0 = x.ld; p.ld1 = x1 ? x1->ld : 0;
            p.B = x.B; p.Hs = x.H; p.Ws = x.W; p.up = up; p.Ho = y.H; p.Wo = y.W; p.KH = w.KH; p.KW = w.KW;
            p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.Cout = w.Cout; p.ldy = y.ld; p.act = act;
            if (res) { ex.err = -3; g_err = "direct conv has no residual path"; return; }
            ex.check(rs_direct_conv_launch(&p, x.dt, y.dt, ex.st), "direct_conv");
        } else {
            IGemmParams p = conv_params(w, x, x1, y, stride, pad_t, pad_l, up, act, res, out_scale);
            p.no_halo = (x.dt == RS_F16S && big(w)) ? 1 : 0;   // (the launcher picks the kernel from the parameter block: tell it what halo_conv() decided)
            p.splitk = splitk; p.partial = partial;
            p.xcoef = xcoef; p.xact = xact;
            if (skw) { p.sx = skx->p; p.sw = skw->w_for(x.dt); p.sbias = skw->bias; p.sC = skx->C; p.sld = skx->ld; p.ww = nullptr; }   // folded 1x1 shortcut (skip_fold()): the halo kernel's
            if (y.st) {   // statistics for the consuming GroupNorm: the halo kernel's or the generic split kernel's epilogue (or their split-K reduce)
                const bool halo = !x1 && w.KH == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && up == 1 && halo_conv(w, x, y, res, nullptr, nullptr, skw != nullptr);
                if (halo || (!x1 && x.dt == RS_F16S && y.dt == RS_F16S)) { p.ystats = y.st; p.ystats_ld = y.stld; }
                else { ex.err = -3; g_err = "output statistics requested from a conv whose kernel cannot produce them"; return; }
                (void)ex.fill_tail(y.st_prod, y.B, p.tail);   // ... and that GroupNorm's coefficients too (gn_tail.h)
            }
            if (!p.w) { ex.err = -3; g_err = "weights for this precision were not packed (enable_f16/enable_f32/enable_split)"; return; }
            ex.igemm(p, x.dt, y.dt, 1, "igemm");
        }
    }
    void zero(Exec& ex, const View& v) {
        if (ex.dry) return;
        const hipError_t e = hipMemsetAsync(v.p, 0, (size_t)v.B * v.H * v.W * v.ld * rs_dtype_size(v.dt), ex.st);
        ex.check(e == hipSuccess ? 0 : -1, "memset");
    }
    void conv3(Exec& ex, const ConvW& w, const View& x, const View& y, const View* res = nullptr, int act = 0, const float* xcoef = nullptr,
               int xact = RS_ACT_NONE, const ConvW* skw = nullptr, const View* skx = nullptr) {
        conv(ex, w, x, nullptr, y, 1, 1, 1, 1, act, res, 1.f, xcoef, xact, skw, skx);
    }
    // A ResBlock's 1x1 shortcut (models/unet.py:178-183,205-206; ldm/modules/diffusionmodules/model.py:121-127,148-149) as extra K columns
    // of its second 3x3 conv (IGemmParams::sx) instead of a GEMM launch + a tensor + a residual read: split storage on the halo kernel's
    // 8-wave big-plane tiles, whole 32-channel chunks of the block input, no weight that needs the unscaled path.  RS_SKIP_FOLD=0: off.
    bool skip_fold(const Exec& ex, const ResBlockW& r, const View& X, const View& h1, const View& Y) const {
        static const bool on = []() { const char* e = getenv("RS_SKIP_FOLD"); return !(e && e[0] == '0'); }();
        static const bool fold = []() { const char* e = getenv("RS_GN_CONV_FOLD"); return !(e && e[0] == '0'); }();
        static const bool on16 = []() { const char* e = getenv("RS_SKIP_FOLD_F16"); return !(e && e[0] == '0'); }();   // (fp16 storage: the decoder's nin_shortcuts, the fp16 policy's UNet)
        if (!on || !fold || ex.dbg || !r.has_skip || (X.dt != RS_F16S && !(X.dt == RS_F16 && on16)) || Y.dt != X.dt) return false;
        if (X.dt == RS_F16S && (big(r.skip) || big(r.c2))) return false;
        if (!r.skip.w_for(X.dt) || r.skip.KH != 1 || X.C != r.skip.CinP || (X.C % 32) || (X.ld % 8) || X.H != Y.H || X.W != Y.W) return false;
        int sk = 1, seg = 0;
        return halo_conv(r.c2, h1, Y, nullptr, &sk, &seg, true) && sk == 1 && seg == 0;   // (the HALO kernel's plan: the fold - a GEMM launch saved - beats the Winograd kernel's 10 %)
    }
    // Attach a statistics buffer to a tensor that is about to be produced by conv `w` from `x` (+res) IF its kernel can leave them: the halo
    // kernel (one partial set per 256- or 128-pixel tile of one image), the generic split-storage kernel (RS_GN_GEN_STATS, default on: one
    // set per 128- / 64-pixel tile), or - split-K launches of either - the reduce kernel (slabs of 256 pixels / the whole small image).
    // The buffer lives in the coefficient pool (reset per network body), so a block's output may carry it to whoever consumes it later
    // (the next block, the decoder's concatenation).
    void want_stats(Exec& ex, const ConvW& w, const View& x, View& y, const View* res, int stride = 1, int pad = 1, int up = 1, bool folded = false) {
        static const bool on = []() { const char* e = getenv("RS_GN_EPI_STATS"); return !(e && e[0] == '0'); }();
        static const bool gen = []() { const char* e = getenv("RS_GN_GEN_STATS"); return !(e && e[0] == '0'); }();
        const int HW = y.H * y.W;
        y.st = nullptr; y.st2 = nullptr; y.st_prod = -1;
        if (!on || ex.dbg || w.direct) return;
        const IGemmParams pp = conv_params(w, x, nullptr, y, stride, pad, pad, up, 0, res, 1.f);
        int spx = 0;
        bool wino = false;
        if (w.KH == 3 && stride == 1 && pad == 1 && up == 1 && halo_conv(w, x, y, res, nullptr, nullptr, folded, &wino)) spx = wino ? rs_wino_stats_px() : rs_igemm4_stats_px(&pp, x.dt);
        else if (gen && x.dt == RS_F16S && y.dt == RS_F16S && x.C == w.CinP)
            spx = rs_igemm_split_stats_px(&pp, rs_igemm_splitk_plan(pp.M, w.Cout, w.KH * w.KW * x.C, x.dt));
        if (spx <= 0 || (HW % spx)) return;
        y.stS = HW / spx; y.stld = y.C;
        y.st = ex.pool((size_t)y.B * y.stS * y.stld * 2 * sizeof(float));
        y.st_prod = ex.prod_seq++;
    }
    // GroupNorm (+FiLM) + SiLU + 3x3 conv (models/unet.py:128-147,198-203; ldm/modules/diffusionmodules/model.py:129-147): on the
    // halo kernel the GroupNorm only produces per-(image, channel) affine coefficients and the conv applies them to the RAW tensor
    // in LDS (bit-identical to normalising first); otherwise normalise into a scratch tensor and convolve that
    void gn_silu_conv3(Exec& ex, const GNW& g, const ConvW& w, const View& X, const View& Y, float eps, const float* film, const View* res,
                       const ConvW* skw = nullptr, const View* skx = nullptr) {
        static const bool fold = []() { const char* e = getenv("RS_GN_CONV_FOLD"); return !(e && e[0] == '0'); }();
        if (fold && !ex.dbg && halo_conv(w, X, Y, res, nullptr, nullptr, skw != nullptr)) {
            const float* coef = gn_coef(ex, g, X, eps, film);
            conv3(ex, w, X, Y, res, 0, coef, RS_ACT_SILU, skw, skx);
            return;
        }
        if (skw) { if (!ex.err) { ex.err = -3; g_err = "folded shortcut planned for a conv that does not run on the halo kernel"; } return; }
        View t = ex.T(X.B, X.H, X.W, X.C, X.dt);
        gn(ex, g, X, t, eps, RS_ACT_SILU, film);
        conv3(ex, w, t, Y, res);
    }
    // GroupNorm + SiLU + conv3x3 to <= 4 channels (the UNet's `out`, Encoder / Decoder norm_out + conv_out) -> fp32 NHWC `o`: one fused pass
    // (gn_silu_head_conv_kernel) over the raw tensor with the GroupNorm as coefficients; RS_HEAD_FUSED=0 or a debug trace: normalise, then
    // the implicit-GEMM conv (the round-3 path)
    void head(Exec& ex, const GNW& g, const ConvW& w, const View& X, const View& o, float eps) {
        static const bool fused = []() { const char* e = getenv("RS_HEAD_FUSED"); return !(e && e[0] == '0'); }();
        if (fused && !ex.dbg && w.wd && !w.direct && o.dt == RS_F32 && X.C == w.Cin && (X.C % 8) == 0 && (X.ld % 8) == 0 && w.Cout <= 4) {
            const float* coef = gn_coef(ex, g, X, eps, nullptr);
            if (!ex.dry) ex.check(rs_head_conv_launch(X.p, X.dt, coef, w.wd, w.bias, (float*)o.p, X.B, X.H, X.W, X.C, X.ld, w.Cout, o.ld, ex.st), "head conv");
            return;
        }
        const size_t mk = ex.mark();
        View t = ex.T(X.B, X.H, X.W, X.C, X.dt);
        gn(ex, g, X, t, eps, RS_ACT_SILU);
        conv(ex, w, t, nullptr, o, 1, 1, 1, 1, 0, nullptr);
        ex.reset(mk);
    }
    void conv1(Exec& ex, const ConvW& w, const View& x, const View& y, const View* res = nullptr, int act = 0) {
        conv(ex, w, x, nullptr, y, 1, 0, 0, 1, act, res);
    }
    // The per-(image, channel) affine [B][2][C] of a GroupNorm whose consumer applies it while loading x (halo conv, fused Swin kernels).
    // Three ways to get it, cheapest first: the launch that produced x's statistics wrote it already (tail, planned by the dry pass);
    // ONE statistics launch whose last workgroup per image writes it (x has no producer statistics); or the coefficient kernel over the
    // producer's partials (RS_GN_TAIL=0, or a producer that is claimed by another GroupNorm).
    const float* gn_coef(Exec& ex, const GNW& g, const View& x, float eps, const float* film) {
        static const bool tails = []() { const char* e = getenv("RS_GN_TAIL"); return !(e && e[0] == '0'); }();
        float* coef = ex.pool((size_t)x.B * 2 * x.C * sizeof(float));
        const int me = ex.gn_seq++;
        if (tails && !ex.dbg && x.st && x.st_prod >= 0 && ex.plan && x.C <= 1280) {
            std::vector<TailPlan>& pl = *ex.plan;
            if (ex.dry) {
                if ((int)pl.size() <= x.st_prod) pl.resize(x.st_prod + 1);
                TailPlan& t = pl[x.st_prod];
                if (!t.on) {
                    t.on = true; t.consumer = me; t.gamma = g.gamma; t.beta = g.beta; t.film = film; t.eps = eps; t.C = x.C; t.HW = x.H * x.W;
                    t.coef_off = (size_t)((char*)coef - ex.pool_base);
                    if (x.st2) { t.two = true; t.st2_off = (size_t)((char*)x.st2 - ex.pool_base); t.st2S = x.st2S; t.st2ld = x.st2ld; }
                    ex.ticket_used += (size_t)x.B;   // (drawn by the producer's launch in the real pass)
                }
            }
            const TailPlan* t = ex.tail_of(x.st_prod);
            if (t && t->consumer == me) return coef;   // written by the producer's tail: nothing to launch
        }
        gn(ex, g, x, x, eps, RS_ACT_NONE, film, coef);
        return coef;
    }
    // `coef` non-null: statistics + affine coefficients only ([B][2][C] floats), y is not written (fused consumer kernels)
    void gn(Exec& ex, const GNW& g, const View& x, const View& y, float eps, int act, const float* film = nullptr, float* coef = nullptr) {
        const int HW = x.H * x.W;
        // slab counts: enough workgroups to fill the chip, but every apply workgroup re-derives the per-channel
        // coefficients, so slabs must stay fat enough to amortise that (RS_GN_BLOCKS_* override the targets for tuning)
        static const int tgt1 = []() { const char* e = getenv("RS_GN_BLOCKS_STATS"); return e ? atoi(e) : 1024; }();
        static const int tgt2 = []() { const char* e = getenv("RS_GN_BLOCKS_APPLY"); return e ? atoi(e) : 2048; }();
        static const int minpx = []() { const char* e = getenv("RS_GN_MIN_PIXELS"); return e ? atoi(e) : 8; }();
        int S = std::max(1, std::min(64, tgt1 / std::max(1, x.B)));
        S = std::max(1, std::min(S, HW / minpx));
        int S2 = std::max(1, std::min(HW / minpx, std::max(1, tgt2 / std::max(1, x.B))));
        float* partial = (float*)ex.raw((size_t)x.B * S * 32 * 2 * sizeof(float));
        // coefficient-only without producer statistics: one launch, the statistics kernel's last workgroup per image finishes (gn_tail.h)
        const bool have_cp = x.st != nullptr;   // per-channel partials from the producer(s): no statistics pass
        unsigned* ticket = (coef && !have_cp && HW > 256) ? ex.tickets(x.B) : nullptr;
        if (ex.dry) return;
        GNParams p{};
        p.ticket = ticket;
        p.x = x.p; p.y = y.p; p.gamma = g.gamma; p.beta = g.beta; p.film = film; p.partial = partial;
        p.B = x.B; p.HW = HW; p.C = x.C; p.ldx = x.ld; p.ldy = y.ld; p.S = S; p.groups = 32; p.eps = eps; p.act = act; p.coef = coef;
        if (x.st) { p.cpartial = x.st; p.cp_ld = x.stld; p.S = x.stS; }   // per-channel partials from the producing conv: no statistics pass
        if (x.st && x.st2) { p.cpartial2 = x.st2; p.cp2_ld = x.st2ld; p.cp2_S = x.st2S; p.cp_n0 = x.st_n0; }   // ... of both halves of a concatenation
        // algorithmic bytes: normalise = read once + write once; coefficients only = read once, or nothing when the statistics
        // come from the producing conv's epilogue
        ex.gn_bytes += (double)x.B * HW * x.C * (double)rs_dtype_size(x.dt) * (coef ? (have_cp ? 0.0 : 1.0) : 2.0);
        ++ex.gn_launches;
        hipEvent_t e0, e1;
        Exec::bracket(ex.prof_gn, ex.st, e0, e1);
        const int nk = rs_groupnorm_launch(&p, x.dt, S2, ex.st);   // number of kernels launched, or < 0
        ex.check(nk < 0 ? nk : 0, "groupnorm");
        if (nk > 1) ex.launches += nk - 1;
        if (e1) (void)hipEventRecord(e1, ex.st);
    }
    // models/unet.py:186-206 (use_scale_shift_norm path); eps 1e-5 (basic_ops.py:96 default GroupNorm eps)
    // `out_stats`: conv2's epilogue also leaves the statistics (and, when the dry pass planned it, the coefficients) for the GroupNorm that
    // consumes Y - the next block's in_layers[0], possibly through the decoder's concatenation
    void resblock(Exec& ex, const ResBlockW& r, const View& X, View& Y, const float* film_row, bool out_stats = true) {
        const size_t mk = ex.mark();
        View h1 = ex.T(X.B, X.H, X.W, r.Cout, X.dt);
        want_stats(ex, r.c1, X, h1, nullptr);   // conv1's epilogue leaves the statistics norm2 needs
        gn_silu_conv3(ex, r.n1, r.c1, X, h1, 1e-5f, nullptr, nullptr);
        ex.tr("conv1", h1);
        const float* film = film_row ? film_row + r.film_off : nullptr;
        if (skip_fold(ex, r, X, h1, Y)) {
            if (out_stats) want_stats(ex, r.c2, h1, Y, nullptr, 1, 1, 1, true);
            gn_silu_conv3(ex, r.n2, r.c2, h1, Y, 1e-5f, film, nullptr, &r.skip, &X);
        } else if (r.has_skip) {
            View sk = ex.T(X.B, X.H, X.W, r.Cout, X.dt);
            conv1(ex, r.skip, X, sk);
            if (out_stats) want_stats(ex, r.c2, h1, Y, &sk);
            gn_silu_conv3(ex, r.n2, r.c2, h1, Y, 1e-5f, film, &sk);
        } else {
            if (out_stats) want_stats(ex, r.c2, h1, Y, &X);
            gn_silu_conv3(ex, r.n2, r.c2, h1, Y, 1e-5f, film, &X);
        }
        ex.reset(mk);
    }
    // ldm/modules/diffusionmodules/model.py:129-149 (temb=None), GroupNorm eps 1e-6 (model.py:46-47)
    void resnet(Exec& ex, const ResBlockW& r, const View& X, View& Y, bool out_stats = true) {
        const size_t mk = ex.mark();
        View h1 = ex.T(X.B, X.H, X.W, r.Cout, X.dt);
        want_stats(ex, r.c1, X, h1, nullptr);
        gn_silu_conv3(ex, r.n1, r.c1, X, h1, 1e-6f, nullptr, nullptr);
        if (skip_fold(ex, r, X, h1, Y)) {
            if (out_stats) want_stats(ex, r.c2, h1, Y, nullptr, 1, 1, 1, true);
            gn_silu_conv3(ex, r.n2, r.c2, h1, Y, 1e-6f, nullptr, nullptr, &r.skip, &X);
        } else if (r.has_skip) {
            View sk = ex.T(X.B, X.H, X.W, r.Cout, X.dt);
            conv1(ex, r.skip, X, sk);
            if (out_stats) want_stats(ex, r.c2, h1, Y, &sk);
            gn_silu_conv3(ex, r.n2, r.c2, h1, Y, 1e-6f, nullptr, &sk);
        } else {
            if (out_stats) want_stats(ex, r.c2, h1, Y, &X);
            gn_silu_conv3(ex, r.n2, r.c2, h1, Y, 1e-6f, nullptr, &X);
        }
        ex.reset(mk);
    }
    // models/swin_transformer.py:427-442 with the two SwinTransformerBlocks (:238-281) inlined
    void basiclayer(Exec& ex, const BasicLayerW& b, const View& X, View& Y, bool out_stats = true) {
        const size_t mk = ex.mark();
        const int E = b.E, heads = cfg.unet.num_heads;
        View e = ex.T(X.B, X.H, X.W, E, X.dt);
        want_stats(ex, b.embed, X, e, nullptr, 1, 0, 1);   // patch_embed's epilogue: statistics (+ coefficients) for the first block's norm1
        conv1(ex, b.embed, X, e);
        ex.tr("embed", e);
        int bi = 0;
        for (const SwinBlockW& s : b.blocks) {
            const std::string bp = "blk" + std::to_string(bi++) + ".";
            // fp16 storage: qkv projection fused into the attention kernel (the [M][3E] tensor never reaches HBM); RS_ATTN_FUSED=0
            // and the debug trace (which records qkv) keep the two launches.  RS_GN_FOLD (default on): the fused consumer kernels
            // also apply the GroupNorm affine while they load their input, so norm1 / norm2 only produce [B][2][E] coefficients.
            static const int attn_fused = []() { const char* v = getenv("RS_ATTN_FUSED"); return v ? atoi(v) : 2; }();
            static const int gn_fold = []() { const char* v = getenv("RS_GN_FOLD"); return v ? atoi(v) : 1; }();
            // split storage: the same fusion in win_attn_split.hip (RS_ATTN_FUSED_SPLIT=0: separate qkv GEMM, attention, projection GEMM)
            static const int attn_fused_split = []() { const char* v = getenv("RS_ATTN_FUSED_SPLIT"); return v ? atoi(v) : 1; }();
            const bool big_attn = X.dt == RS_F16S && (big(s.qkv) || big(s.proj)), big_mlp = X.dt == RS_F16S && (big(s.fc1) || big(s.fc2));   // (|w| >= 30: unfused path)
            const bool fuse_qkv = rs_win_attn_qkv_supported(heads, E) && s.bias_n && !ex.dbg && !big_attn &&
                                  ((attn_fused && X.dt == RS_F16 && s.qkv.wh_frag) || (attn_fused_split && X.dt == RS_F16S && s.qkv.ws_frag && s.proj.ws_frag));
            const bool fold1 = fuse_qkv && gn_fold;
            View n;
            const float* coef1 = nullptr;
            if (fold1) coef1 = gn_coef(ex, s.n1, e, 1e-5f, nullptr);
            else { n = ex.T(X.B, X.H, X.W, E, X.dt); gn(ex, s.n1, e, n, 1e-5f, RS_ACT_NONE); }
            View qkv;
            if (!fuse_qkv) {
                qkv = ex.T(X.B, X.H, X.W, 3 * E, X.dt);
                conv1(ex, s.qkv, n, qkv);
                ex.tr(bp + "qkv", qkv);
            }
            const bool fuse_proj = fuse_qkv && (X.dt == RS_F16S || (attn_fused >= 2 && s.proj.wh_frag));   // ... and the output projection + shortcut as well
            View a, e2;
            if (fuse_proj) e2 = ex.T(X.B, X.H, X.W, E, X.dt); else a = ex.T(X.B, X.H, X.W, E, X.dt);
            // fp16 storage: one fused MLP launch, the [M][4E] hidden tensor never reaches HBM (swin_mlp.hip); RS_MLP_FUSED=0 or a
            // small token count (RS_MLP_FUSED_MINM) keep the two GEMMs
            static const int mlp_fused = []() { const char* v = getenv("RS_MLP_FUSED"); return v ? atoi(v) : 3; }();   // bit 0: fp16, bit 1: split storage
            static const int mlp_minm = []() { const char* v = getenv("RS_MLP_FUSED_MINM"); return v ? atoi(v) : 16384; }();   // (8192 tokens: 32 us fused vs 14 + 15 us apart)
            const int Mtok = X.B * X.H * X.W, HWt = X.H * X.W;
            const bool fuse_mlp = mlp_fused && !big_mlp && (X.dt == RS_F16 || (X.dt == RS_F16S && (mlp_fused & 2))) && rs_swin_mlp_supported(E, s.fc1.Cout) &&
                                  s.fc2.Cout == E && Mtok >= mlp_minm && s.fc1.w_for(X.dt) && s.fc2.w_for(X.dt);
            const bool fold2 = fuse_mlp && gn_fold && !ex.dbg && HWt % 128 == 0;
            // The fused kernels' epilogues leave the statistics of their outputs for the GroupNorm that reads them (attention -> norm2: one
            // partial set per window; MLP -> the next block's norm1: one per 128-token tile) and - round 4 - the launch that completes an
            // image's statistics writes that GroupNorm's coefficients (tail): no pass over those tensors, no coefficient launch.
            //   split storage: default on (RS_GN_SWIN_STATS_SPLIT=0: statistics pass + tail as for any tensor without producer statistics);
            //   fp16 storage: never (round 3's form - statistics only, coefficient kernel - measured slower than the passes it removes,
            //   profiles/r3_negative_results.txt; its knob left the tree in round 5).
            static const bool swin_stats_split = []() { const char* v = getenv("RS_GN_SWIN_STATS_SPLIT"); return !(v && v[0] == '0'); }();
            const bool sstats = X.dt == RS_F16S && swin_stats_split && !ex.dbg;
            if (fuse_proj && sstats && fold2) {   // (only where norm2 is a coefficient-only GroupNorm: the fused MLP applies it)
                e2.stS = (X.H / 8) * (X.W / 8); e2.stld = E;
                e2.st = ex.pool((size_t)X.B * e2.stS * e2.stld * 2 * sizeof(float));
                e2.st_prod = ex.prod_seq++;
            }
            if (fuse_qkv) {
                if (!ex.dry) {
                    WinAttnParams p{};
                    p.bias_n = s.bias_n; p.bias_c = s.bias_c; p.B = X.B; p.H = X.H; p.W = X.W; p.heads = heads; p.shift = s.shift;
                    p.scale = 1.0f / std::sqrt((float)(E / heads));
                    if (fold1) { p.x = e.p; p.ldx = e.ld; p.xcoef = coef1; } else { p.x = n.p; p.ldx = n.ld; }
                    p.wqkv = s.qkv.w_frag_for(X.dt); p.bqkv = s.qkv.bias;
                    if (fuse_proj) { p.out = e2.p; p.ldo = e2.ld; p.wproj = s.proj.w_frag_for(X.dt); p.bproj = s.proj.bias; p.res = e.p; p.ldres = e.ld;
                                     p.ystats = e2.st; p.ystats_ld = e2.stld;
                                     if (e2.st) (void)ex.fill_tail(e2.st_prod, X.B, p.tail); }
                    else { p.out = a.p; p.ldo = a.ld; }
                    ex.win_attn_qkv(p, E, X.dt);
                }
            } else if (!ex.dry) {
                WinAttnParams p{};
                p.qkv = qkv.p; p.out = a.p; p.bias_t = s.bias_t; p.bias_n = s.bias_n; p.B = X.B; p.H = X.H; p.W = X.W; p.heads = heads;
                p.shift = s.shift; p.ldq = qkv.ld; p.ldo = a.ld; p.scale = 1.0f / std::sqrt((float)(E / heads));
                ex.check(rs_win_attn_launch(&p, X.dt, ex.st), "win_attn");
            }
            if (!fuse_proj) {
                ex.tr(bp + "attn", a);
                e2 = ex.T(X.B, X.H, X.W, E, X.dt);
                conv1(ex, s.proj, a, e2, &e);
            }
            ex.tr(bp + "proj", e2);
            View e3;
            View n2;
            const float* coef2 = nullptr;
            if (fold2) coef2 = gn_coef(ex, s.n2, e2, 1e-5f, nullptr);
            else { n2 = ex.T(X.B, X.H, X.W, E, X.dt); gn(ex, s.n2, e2, n2, 1e-5f, RS_ACT_NONE); }
            // patch_unembed inside the last block's MLP launch (swin_mlp.hip, NO != E): split storage, norm2 folded (the kernel's x is the block's
            // raw input = its shortcut), the product matrix below the scaled-fragment limit.  RS_UNEMBED_FOLD=0: the 1x1 conv as a launch.
            static const bool unfold_on = []() { const char* v = getenv("RS_UNEMBED_FOLD"); return !(v && v[0] == '0'); }();
            const bool unfold = unfold_on && fuse_mlp && fold2 && X.dt == RS_F16S && &s == &b.blocks.back() && b.has_unfold && b.unfold.ws &&
                                !big(b.unfold) && Y.dt == RS_F16S && Y.C == b.C && rs_swin_mlp_split_unembed_supported(E, s.fc1.Cout, b.C) &&
                                (Y.ld % 8) == 0 && (e2.ld % 8) == 0;   // (the launcher's alignment preconditions: an unaligned view falls back to the separate 1x1 conv instead of failing the pass - ADVICE r5)
            if (unfold) {
                Y.st = nullptr; Y.st2 = nullptr; Y.st_prod = -1;
                if (out_stats && sstats && HWt % 128 == 0 && Mtok % 128 == 0) {   // statistics (+ tail) for the GroupNorm that consumes the layer's output
                    Y.stS = HWt / 128; Y.stld = Y.C;
                    Y.st = ex.pool((size_t)X.B * Y.stS * Y.stld * 2 * sizeof(float));
                    Y.st_prod = ex.prod_seq++;
                }
                if (!ex.dry) {
                    GNTail tl{};
                    const GNTail* tlp = (Y.st && ex.fill_tail(Y.st_prod, X.B, tl)) ? &tl : nullptr;
                    ex.swin_mlp(e2.p, s.fc1.w_for(X.dt), s.fc1.bias, b.unfold.ws, b.unfold.bias, nullptr, Y.p, Mtok, e2.ld, 0, Y.ld, E, s.fc1.Cout, coef2, HWt,
                                X.dt, Y.st, Y.stld, tlp, b.C);
                }
                ex.reset(mk);
                return;
            }
            if (fuse_mlp) {
                e3 = ex.T(X.B, X.H, X.W, E, X.dt);
                const bool is_last = &s == &b.blocks.back();   // (the last block's output feeds patch_unembed, not a GroupNorm)
                if (sstats && !is_last && HWt % 128 == 0 && Mtok % 128 == 0) {
                    e3.stS = HWt / 128; e3.stld = E;
                    e3.st = ex.pool((size_t)X.B * e3.stS * e3.stld * 2 * sizeof(float));
                    e3.st_prod = ex.prod_seq++;
                }
                if (!ex.dry) {
                    const void* w1 = s.fc1.w_for(X.dt); const void* w2 = s.fc2.w_for(X.dt);
                    GNTail tl{};
                    const GNTail* tlp = (e3.st && ex.fill_tail(e3.st_prod, X.B, tl)) ? &tl : nullptr;
                    if (fold2) ex.swin_mlp(e2.p, w1, s.fc1.bias, w2, s.fc2.bias, e2.p, e3.p, Mtok, e2.ld, e2.ld, e3.ld, E, s.fc1.Cout, coef2, HWt, X.dt, e3.st, e3.stld, tlp);
                    else ex.swin_mlp(n2.p, w1, s.fc1.bias, w2, s.fc2.bias, e2.p, e3.p, Mtok, n2.ld, e2.ld, e3.ld, E, s.fc1.Cout, nullptr, HWt, X.dt, e3.st, e3.stld, tlp);
                }
            } else {
                View f = ex.T(X.B, X.H, X.W, s.fc1.Cout, X.dt);
                conv1(ex, s.fc1, n2, f, nullptr, RS_ACT_GELU);
                e3 = ex.T(X.B, X.H, X.W, E, X.dt);
                conv1(ex, s.fc2, f, e3, &e2);
            }
            ex.tr(bp + "out", e3);
            e = e3;
        }
        if (out_stats) want_stats(ex, b.unembed, e, Y, nullptr, 1, 0, 1);
        conv1(ex, b.unembed, e, Y);
        ex.reset(mk);
    }
    // model.py:179-203: x + proj_out(softmax(q k^T / sqrt(C)) v); S is materialised in fp32 per image chunk
    void attnblock(Exec& ex, const AttnW& a, const View& X, View& Y) {
        const size_t mk = ex.mark();
        const int C = a.C, T = X.H * X.W, dt = X.dt;
        View n = ex.T(X.B, X.H, X.W, C, dt);
        gn(ex, a.norm, X, n, 1e-6f, RS_ACT_NONE);
        View q = ex.T(X.B, X.H, X.W, C, dt), k = ex.T(X.B, X.H, X.W, C, dt);
        conv1(ex, a.q, n, q);
        conv1(ex, a.k, n, k);
        View o = ex.T(X.B, X.H, X.W, C, dt);
        // fp16 storage: streaming attention (ae_attn.hip) - S never reaches HBM; RS_AE_FLASH=0 keeps the row-block path below (A/B runs),
        // which also serves fp32 / split storage and token counts that are not multiples of 128
        static const bool flash_on = []() { const char* e = getenv("RS_AE_FLASH"); return !(e && e[0] == '0'); }();
        // split storage (the encoder of the parity policy): the same on (hi, lo) pairs, ae_attn_split.hip (round 4)
        const bool flash16 = dt == RS_F16 && rs_ae_flash_supported(C, T) && a.v.wh;
        const bool flash_s = dt == RS_F16S && rs_ae_flash_split_supported(C, T) && a.v.ws;
        if (flash_on && (flash16 || flash_s)) {
            char* vTa = (char*)ex.raw((size_t)X.B * C * T * rs_dtype_size(dt));
            if (!ex.dry) {
                // vT[z][c][t] = sum_k Wv[c][k] n[z][t][k]   (bias added to the attention output: softmax rows sum to 1)
                gemm_nt(ex, a.v.w_for(dt), 0, n.p, (long long)T * C, nullptr, vTa, (long long)C * T, X.B, C, T, C, 1.f, dt, dt);
                ex.ae_flash(q.p, q.ld, k.p, k.ld, vTa, a.v.bias, o.p, o.ld, X.B, T, C, 1.0f / std::sqrt((float)C), dt);
            }
            want_stats(ex, a.proj, o, Y, &X, 1, 0, 1);
            conv1(ex, a.proj, o, Y, &X);
            ex.reset(mk);
            return;
        }
        // The score matrix is materialised, but never more than `budget` floats of it at a time (4 GiB of fp32 S + the same
        // number of P elements): several images per pass while a whole T x T matrix fits (the 64 x 64 latents of the shipped
        // configs: T = 4096), otherwise one image in blocks of query rows (softmax is row-wise, so row blocks are independent).
        // That is what lets the tiled path run the reference's real tile sizes (inference_resshift.py:149-161): a 256 x 256 LR
        // tile is T = 65 536 tokens, 512 x 512 T = 262 144.  At d = 512 a flash-style kernel would have to stream a 64 KB K
        // tile AND a 64 KB V tile through LDS per 64 keys (one ds_read_b128 per MFMA: LDS-bound); the two GEMMs + row softmax
        // run on the tuned implicit-GEMM path instead and their scratch is bounded here.
        static const size_t budget = []() { const char* e = getenv("RS_ATTN_S_FLOATS"); return e ? (size_t)atoll(e) : ((size_t)1 << 30); }();
        const size_t tt = (size_t)T * T;
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)X.B, budget / tt));
        int rows = T;
        if (tt > budget) rows = (int)std::max<size_t>(128, std::min<size_t>((size_t)T, (budget / (size_t)T) / 128 * 128));
        const size_t es = rs_dtype_size(dt);
        char* vT = (char*)ex.raw((size_t)chunk * C * T * es);
        float* S = (float*)ex.raw((size_t)chunk * rows * T * sizeof(float));
        char* P = (char*)ex.raw((size_t)chunk * rows * T * es);
        if (!ex.dry) {
            for (int b0 = 0; b0 < X.B; b0 += chunk) {
                const int nz = std::min(chunk, X.B - b0);
                const size_t boff = (size_t)b0 * T * C * es;
                // vT[z][c][t] = sum_k Wv[c][k] n[z][t][k]   (bias folded into the PV epilogue: softmax rows sum to 1)
                gemm_nt(ex, a.v.w_for(dt), 0, (char*)n.p + boff, (long long)T * C, nullptr, vT, (long long)C * T, nz, C, T, C, 1.f, dt, dt);
                for (int r0 = 0; r0 < T; r0 += rows) {
                    const int nr = std::min(rows, T - r0);   // (rows == T unless a single image is processed in row blocks)
                    const size_t roff = boff + (size_t)r0 * C * es;
                    gemm_nt(ex, (char*)q.p + roff, (long long)T * C, (char*)k.p + boff, (long long)T * C, nullptr, S, (long long)nr * T, nz, nr, T, C,
                            1.0f / std::sqrt((float)C), dt, RS_F32);
                    ex.check(rs_softmax_rows_launch(S, P, dt, (long long)nz * nr, T, T, T, ex.st), "softmax");
                    gemm_nt(ex, P, (long long)nr * T, vT, (long long)C * T, a.v.bias, (char*)o.p + roff, (long long)T * C, nz, nr, C, T, 1.f, dt, dt);
                }
            }
        }
        want_stats(ex, a.proj, o, Y, &X, 1, 0, 1);
        conv1(ex, a.proj, o, Y, &X);
        ex.reset(mk);
    }
    // y[z][m][n] = scale * sum_k A[z][m][k] B[z][n][k] (+bias[n])
    void gemm_nt(Exec& ex, const void* A, long long bsA, const void* Bm, long long bsB, const float* bias, void* y, long long bsY, int nz,
                 int M, int N, int K, float scale, int in_dt, int out_dt) {
        if (!A || !Bm) { if (!ex.err) { ex.err = -3; g_err = "gemm operand missing (precision not packed?)"; } return; }
        IGemmParams p{};
        p.x0 = A; p.w = Bm; p.bias = bias; p.y = y; p.C0 = K; p.ld0 = K; p.B = 1; p.Hs = M; p.Ws = 1; p.up = 1; p.Ho = M; p.Wo = 1;
        p.KH = 1; p.KW = 1; p.stride = 1; p.Cout = N; p.ldy = N; p.M = M; p.Ktot = K; p.out_scale = scale;
        p.bs_x0 = bsA; p.bs_w = bsB; p.bs_y = bsY;
        ex.igemm(p, in_dt, out_dt, nz, "gemm_nt");
    }

    // ---------------------------------------------------------------- FiLM cache
    // emb = time_embed(timestep_embedding(t)) (unet.py:874, basic_ops.py:99-117); per ResBlock
    // emb_out = Linear(SiLU(emb)) (unet.py:161-167,195).  Row layout: [film_total], block r at r.film_off.
    const float* film_row(int t, hipStream_t st) {
        auto it = film_cache.find(t);
        if (it != film_cache.end()) return it->second;
        if (rs_fake_device()) { film_cache[t] = (float*)malloc((size_t)film_total * 4); return film_cache[t]; }   // (plumbing check without a GPU, see run())
        const int mc = cfg.unet.model_channels, half = mc / 2, emb_ch = 4 * mc;
        std::vector<float> e0(mc, 0.f);
        for (int k = 0; k < half; ++k) {
            const float freq = expf((float)(-std::log(10000.0)) * (float)k / (float)half);
            const float arg = (float)t * freq;
            e0[k] = cosf(arg); e0[half + k] = sinf(arg);
        }
        float *d0, *d1, *d2, *row;
        if (hipMalloc(&d0, mc * 4) != hipSuccess || hipMalloc(&d1, emb_ch * 4) != hipSuccess || hipMalloc(&d2, emb_ch * 4) != hipSuccess ||
            hipMalloc(&row, (size_t)film_total * 4) != hipSuccess)
            return nullptr;
        (void)hipMemcpyAsync(d0, e0.data(), mc * 4, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);  // e0 is a stack-lifetime host buffer
        rs_small_linear_launch(d0, te0.wd, te0.bias, d1, 1, mc, emb_ch, 0, 1, st);
        rs_small_linear_launch(d1, te2.wd, te2.bias, d2, 1, emb_ch, emb_ch, 0, 0, st);
        for (ResBlockW* r : film_blocks) rs_small_linear_launch(d2, r->emb.wd, r->emb.bias, row + r->film_off, 1, emb_ch, 2 * r->Cout, 1, 0, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(d2);
        film_cache[t] = row;
        return row;
    }
    void collect_film_blocks() {
        film_blocks.clear();
        if (!cfg.has_unet) return;
        for (auto& b : in_blocks) if (b.has_res) film_blocks.push_back(&b.res);
        film_blocks.push_back(&mid_res1); film_blocks.push_back(&mid_res2);
        for (auto& b : out_blocks) if (b.has_res) film_blocks.push_back(&b.res);
    }

    // ---------------------------------------------------------------- UNet forward
    // x: NCHW fp32 [B,Cz,H,W] (already scaled by _scale_input when called through the drop-in API);
    // lq_feat: optional NHWC view of the (feature-extracted) conditioning; out: NCHW fp32.
    void unet_body(Exec& ex, const float* x, float xscale, const View* lq_feat, const float* lq_nchw, const float* mask_nchw, int Hl, int Wl,
                   float* out, int B, int H, int W, int dt, const float* film) {
        ex.enter_part("unet");
        const rs_unet_config& u = cfg.unet;
        const int n_in = (int)in_blocks.size(), n_out = (int)out_blocks.size();
        const size_t mk0 = ex.mark();
        ex.pool_off = 0;   // coefficient pool: every slot is consumed within this forward; the next one may reuse them (stream order)
        auto lvH = [&](int level) { return H >> level; };
        auto lvW = [&](int level) { return W >> level; };
        // zero-copy concat buffers, one per output block
        std::vector<View> cat(n_out);
        for (int j = 0; j < n_out; ++j) {
            const int i = n_in - 1 - j, lvl = in_blocks[i].level;
            cat[j] = ex.T(B, lvH(lvl), lvW(lvl), h_ch[j] + skip_ch[i], dt);
        }
        auto skip_view = [&](int i) { const int j = n_in - 1 - i; return cat[j].slice(h_ch[j], skip_ch[i]); };
        // statistics of the two halves of every concat buffer (want_stats): the decoder side is channels [0, h_ch), the encoder's skip the
        // rest; once both are known the output block's in_layers[0] GroupNorm needs no pass over the tensor, and the launch that writes
        // the decoder half - always the later one - carries its tail
        struct Half { float* st = nullptr; int S = 0, ld = 0, prod = -1; };
        std::vector<Half> cat_lo(n_out), cat_hi(n_out);
        auto note = [](Half& hf, const View& v) { if (v.st && !v.st2) { hf.st = v.st; hf.S = v.stS; hf.ld = v.stld; hf.prod = v.st_prod; } };
        auto cat_view = [&](int j) {
            View X = cat[j];
            if (cat_lo[j].st && cat_hi[j].st) {
                X.st = cat_lo[j].st; X.stS = cat_lo[j].S; X.stld = cat_lo[j].ld; X.st_prod = cat_lo[j].prod;
                X.st2 = cat_hi[j].st; X.st2S = cat_hi[j].S; X.st2ld = cat_hi[j].ld; X.st_n0 = h_ch[j];
            }
            return X;
        };
        // ---- input conv (unet.py:876-886): cat[x, lq(,mask)]
        const int Cz = u.in_channels;
        View h;
        {
            const size_t mk = ex.mark();
            View y0 = skip_view(0);
            if (fe_convs.empty()) {
                const int cl = fe_out_ch;  // 3 or 4 raw conditioning channels at latent resolution
                // cat[x, lq(, mask)] as one channel-padded NHWC tensor in the working precision (6/7 -> 8 channels)
                View in0 = ex.T(B, H, W, in_blocks[0].conv.CinP, dt);
                if (!ex.dry) {
                    zero(ex, in0);
                    ex.check(rs_nchw_to_nhwc_launch(x, in0.p, dt, B, Cz, H * W, in0.ld, 0, xscale, ex.st), "x->nhwc");
                    if (cl) ex.check(rs_nchw_to_nhwc_launch(lq_nchw, in0.p, dt, B, 3, H * W, in0.ld, Cz, 1.f, ex.st), "lq->nhwc");
                    if (cl == 4) ex.check(rs_nchw_to_nhwc_launch(mask_nchw, in0.p, dt, B, 1, H * W, in0.ld, Cz + 3, 1.f, ex.st), "mask->nhwc");
                }
                want_stats(ex, in_blocks[0].conv, in0, y0, nullptr);
                conv(ex, in_blocks[0].conv, in0, nullptr, y0, 1, 1, 1, 1, 0, nullptr);
            } else {
                // conditioning goes through the strided-conv feature extractor (unet.py:693-702); lq_feat precomputed
                View xin = ex.T(B, H, W, Cz, dt);
                if (!ex.dry) ex.check(rs_nchw_to_nhwc_launch(x, xin.p, dt, B, Cz, H * W, xin.ld, 0, xscale, ex.st), "x->nhwc");
                conv(ex, in_blocks[0].conv, xin, lq_feat, y0, 1, 1, 1, 1, 0, nullptr);
            }
            ex.reset(mk);
            h = y0;
            note(cat_hi[n_in - 1], y0);
            ex.tr("in.0", h);
        }
        // ---- input blocks
        for (int i = 1; i < n_in; ++i) {
            const UBlock& b = in_blocks[i];
            const size_t mk = ex.mark();
            View y = skip_view(i);
            if (b.has_down) {
                want_stats(ex, b.conv, h, y, nullptr, 2, 1, 1);
                conv(ex, b.conv, h, nullptr, y, 2, 1, 1, 1, 0, nullptr);
                ex.tr("in." + std::to_string(i), y);
            } else if (b.has_swin) {
                View r = ex.T(B, h.H, h.W, b.out_ch, dt);
                ex.prefix = "in." + std::to_string(i) + ".res.";
                resblock(ex, b.res, h, r, film, /*out_stats=*/false);   // (r feeds patch_embed, not a GroupNorm)
                ex.prefix = "in." + std::to_string(i) + ".swin.";
                basiclayer(ex, b.swin, r, y);
                ex.prefix.clear();
                ex.tr("in." + std::to_string(i) + ".res", r);
                ex.tr("in." + std::to_string(i), y);
            } else {
                ex.prefix = "in." + std::to_string(i) + ".res.";
                resblock(ex, b.res, h, y, film);
                ex.prefix.clear();
                ex.tr("in." + std::to_string(i), y);
            }
            ex.reset(mk);
            h = y;
            note(cat_hi[n_in - 1 - i], y);
        }
        // ---- middle (unet.py:889)
        {
            const size_t mk = ex.mark();
            View r1 = ex.T(B, h.H, h.W, h.C, dt), r2 = ex.T(B, h.H, h.W, h.C, dt);
            resblock(ex, mid_res1, h, r1, film, /*out_stats=*/false);
            ex.tr("mid.res1", r1);
            basiclayer(ex, mid_swin, r1, r2);
            ex.tr("mid.swin", r2);
            View y = cat[0].slice(0, h_ch[0]);
            resblock(ex, mid_res2, r2, y, film);
            note(cat_lo[0], y);
            ex.tr("mid.res2", y);
            ex.reset(mk);
        }
        // ---- output blocks (unet.py:890-892)
        View last;
        for (int j = 0; j < n_out; ++j) {
            const UBlock& b = out_blocks[j];
            const View X = cat_view(j);
            View y;
            if (j + 1 < n_out) {
                y = cat[j + 1].slice(0, h_ch[j + 1]);
            } else {
                last = ex.T(B, X.H, X.W, b.out_ch, dt);  // stays live for the out head
                y = last;
            }
            const size_t mk = ex.mark();
            if (!b.has_swin && !b.has_up) {
                resblock(ex, b.res, X, y, film);
            } else {
                View cur = ex.T(B, X.H, X.W, b.out_ch, dt);
                resblock(ex, b.res, X, cur, film, /*out_stats=*/false);   // (feeds patch_embed or the upsampling conv)
                if (b.has_swin) {
                    if (b.has_up) {
                        View r2 = ex.T(B, X.H, X.W, b.out_ch, dt);
                        basiclayer(ex, b.swin, cur, r2, /*out_stats=*/false);
                        cur = r2;
                    } else {
                        basiclayer(ex, b.swin, cur, y);
                    }
                }
                if (b.has_up) {
                    if (b.has_upf && upfold_ok(ex, b.upf, cur, y)) {
                        upfold_want_stats(ex, b.upf, cur, y);
                        upfold_conv(ex, b.upf, cur, y);
                    } else {
                        want_stats(ex, b.conv, cur, y, nullptr, 1, 1, 2);
                        conv(ex, b.conv, cur, nullptr, y, 1, 1, 1, 2, 0, nullptr);  // nearest x2 folded into the conv's addressing
                    }
                }
            }
            if (j + 1 < n_out) note(cat_lo[j + 1], y); else last = y;
            ex.tr("out." + std::to_string(j), y);
            ex.reset(mk);
        }
        // ---- out head (unet.py:893-894)
        {
            View o = ex.T(B, H, W, u.out_channels, RS_F32);
            head(ex, out_norm, out_conv, last, o, 1e-5f);
            if (!ex.dry) ex.check(rs_nhwc_to_nchw_launch(o.p, RS_F32, out, B, u.out_channels, H * W, o.ld, 0, ex.st), "out->nchw");
        }
        ex.reset(mk0);
    }
    // feature_extractor(cat[lq, mask]) (unet.py:876-881, 693-702): Conv3x3 -> SiLU -> Downsample conv s2
    View feature_extract(Exec& ex, const float* lq, const float* mask, int B, int Hl, int Wl, int dt) {
        const int cin = cfg.unet.cond_mask ? 4 : 3;
        View cur = ex.T(B, Hl, Wl, fe_convs[0].CinP, dt);  // 3/4 -> 8 zero-padded channels
        if (!ex.dry) {
            zero(ex, cur);
            ex.check(rs_nchw_to_nhwc_launch(lq, cur.p, dt, B, 3, Hl * Wl, cur.ld, 0, 1.f, ex.st), "lq->nhwc");
            if (cin == 4) ex.check(rs_nchw_to_nhwc_launch(mask, cur.p, dt, B, 1, Hl * Wl, cur.ld, 3, 1.f, ex.st), "mask->nhwc");
        }
        for (size_t s = 0; s < fe_convs.size(); ++s) {
            View a = ex.T(B, cur.H, cur.W, fe_convs[s].Cout, dt);
            conv(ex, fe_convs[s], cur, nullptr, a, 1, 1, 1, 1, RS_ACT_SILU, nullptr);
            View d = ex.T(B, cur.H / 2, cur.W / 2, fe_downs[s].Cout, dt);
            conv(ex, fe_downs[s], a, nullptr, d, 2, 1, 1, 1, 0, nullptr);
            cur = d;
        }
        return cur;
    }

    // ---------------------------------------------------------------- AE
    // img NCHW fp32 [B,3,H,W] (or NHWC view if `img_nhwc`) -> z NCHW fp32 [B,embed,H/f,W/f]
    void encode_body(Exec& ex, const View& in_nhwc, float* z_nchw, int dt) {
        ex.enter_part("encoder");
        const rs_ae_config& a = cfg.ae;
        const size_t mk0 = ex.mark();
        ex.pool_off = 0;
        const int B = in_nhwc.B;
        View h = ex.T(B, in_nhwc.H, in_nhwc.W, a.ch, dt);
        want_stats(ex, enc_in, in_nhwc, h, nullptr);
        conv(ex, enc_in, in_nhwc, nullptr, h, 1, 1, 1, 1, 0, nullptr);
        for (int l = 0; l < a.n_levels; ++l) {
            const AELevel& L = enc_levels[l];
            for (const ResBlockW& r : L.blocks) {
                View y = ex.T(B, h.H, h.W, r.Cout, dt);
                resnet(ex, r, h, y);
                h = y;
            }
            if (L.has_resample) {
                // F.pad(x,(0,1,0,1)) + conv stride 2 pad 0 (model.py:80-84)
                View y = ex.T(B, h.H / 2, h.W / 2, h.C, dt);
                want_stats(ex, L.resample, h, y, nullptr, 2, 0, 1);
                conv(ex, L.resample, h, nullptr, y, 2, 0, 0, 1, 0, nullptr);
                h = y;
            }
        }
        View m1 = ex.T(B, h.H, h.W, h.C, dt); resnet(ex, enc_mid1, h, m1);
        View m2 = ex.T(B, h.H, h.W, h.C, dt); attnblock(ex, enc_attn, m1, m2);
        View m3 = ex.T(B, h.H, h.W, h.C, dt); resnet(ex, enc_mid2, m2, m3);
        View zc = ex.T(B, h.H, h.W, a.z_channels, RS_F32);
        head(ex, enc_norm, enc_out, m3, zc, 1e-6f);
        View zq = ex.T(B, h.H, h.W, a.embed_dim, RS_F32);
        conv(ex, quant_conv, zc, nullptr, zq, 1, 0, 0, 1, 0, nullptr);
        if (!ex.dry) ex.check(rs_nhwc_to_nchw_launch(zq.p, RS_F32, z_nchw, B, a.embed_dim, h.H * h.W, zq.ld, 0, ex.st), "z->nchw");
        ex.reset(mk0);
    }
    // z NCHW fp32 [B,embed,h,w] -> img NCHW fp32
    void decode_body(Exec& ex, const float* z_nchw, float zscale, float* img, int32_t* idx_out, int B, int h_, int w_, int force_nq, int dt) {
        ex.enter_part("decoder");
        const rs_ae_config& a = cfg.ae;
        const size_t mk0 = ex.mark();
        ex.pool_off = 0;
        View z = ex.T(B, h_, w_, a.embed_dim, RS_F32);
        if (!ex.dry) ex.check(rs_nchw_to_nhwc_launch(z_nchw, z.p, RS_F32, B, a.embed_dim, h_ * w_, z.ld, 0, zscale, ex.st), "z->nhwc");
        View q = z;
        if (!force_nq) {
            q = ex.T(B, h_, w_, a.embed_dim, RS_F32);
            if (!ex.dry) ex.check(rs_vq_launch((const float*)z.p, codebook, (float*)q.p, idx_out, (long long)B * h_ * w_, a.n_embed, a.embed_dim, ex.st), "vq");
        }
        View pq = ex.T(B, h_, w_, dec_in.CinP, dt);  // z_channels zero-padded to the decoder conv_in's chunk size
        zero(ex, pq);
        conv(ex, post_quant_conv, q, nullptr, pq.slice(0, a.z_channels), 1, 0, 0, 1, 0, nullptr);
        View h = ex.T(B, h_, w_, dec_in.Cout, dt);
        conv(ex, dec_in, pq, nullptr, h, 1, 1, 1, 1, 0, nullptr);
        View m1 = ex.T(B, h.H, h.W, h.C, dt); resnet(ex, dec_mid1, h, m1);
        View m2 = ex.T(B, h.H, h.W, h.C, dt); attnblock(ex, dec_attn, m1, m2);
        View m3 = ex.T(B, h.H, h.W, h.C, dt); resnet(ex, dec_mid2, m2, m3);
        h = m3;
        for (int l = a.n_levels - 1; l >= 0; --l) {
            const AELevel& L = dec_levels[l];
            for (const ResBlockW& r : L.blocks) {
                View y = ex.T(B, h.H, h.W, r.Cout, dt);
                resnet(ex, r, h, y);
                h = y;
            }
            if (L.has_resample) {
                View y = ex.T(B, h.H * 2, h.W * 2, h.C, dt);
                if (L.has_upf && upfold_ok(ex, L.upf, h, y)) upfold_conv(ex, L.upf, h, y);
                else conv(ex, L.resample, h, nullptr, y, 1, 1, 1, 2, 0, nullptr);
                h = y;
            }
        }
        View o = ex.T(B, h.H, h.W, a.out_ch, RS_F32);
        head(ex, dec_norm, dec_out, h, o, 1e-6f);
        if (!ex.dry) ex.check(rs_nhwc_to_nchw_launch(o.p, RS_F32, img, B, a.out_ch, h.H * h.W, o.ld, 0, ex.st), "img->nchw");
        ex.reset(mk0);
    }

    // ---------------------------------------------------------------- run helper (dry sizing pass, then real pass)
    int run(hipStream_t st, const std::function<void(Exec&)>& fn) {
        if (!ready) return fail("weights are not ready (rs_pack_weights / rs_weights_ready not called)");
        Exec d; d.st = st; d.arena = &arena; d.dry = true; d.keep = debug; d.dbg = debug;
        tail_plan.clear();
        d.plan = &tail_plan;
        d.pool_base = (char*)4096;   // never dereferenced; non-null so that "has statistics" (View::st != nullptr) reads the same in both passes
        d.ticket_base = (unsigned*)4096;
        arena.off = 0; arena.peak = 0;
        fn(d);
        if (d.used_split && !(cfg.enable_split && split_ok))
            return fail(!cfg.enable_split ? "split precision requested but the engine was created without enable_split"
                                          : "split precision is not available for these weights: " +
                                                (split_err.empty() ? std::string("a non-finite or out-of-fp16-range conv / linear weight (the packing rank has its name)") : split_err));
        // behind the scratch arena: the GroupNorm coefficient pool and the tails' tickets (gn_tail.h), sized by the dry pass
        const size_t scratch_end = (arena.peak + 255) & ~(size_t)255;
        const size_t pool_bytes = (d.pool_peak + 255) & ~(size_t)255, ticket_bytes = d.ticket_used * sizeof(unsigned);
        const size_t need = scratch_end + pool_bytes + ticket_bytes + 4096;
        // RS_FAKE_DEVICE=1 (plumbing check in a container without a GPU): the scratch arena comes from host memory and every launch simply
        // fails, but the real pass walks its whole control flow - enough to check that it agrees with the dry pass about pools and tickets
        const bool fake = rs_fake_device();
        if (fake && need > arena.cap) { arena.base = (char*)malloc(64); arena.cap = (size_t)1 << 62; }
        if (need > arena.cap) {
            (void)hipStreamSynchronize(st);
            if (arena.base) (void)hipFree(arena.base);
            arena.base = nullptr; arena.cap = 0;
            const size_t want = need + need / 8;
            if (hipMalloc((void**)&arena.base, want) != hipSuccess) return fail("hipMalloc of scratch arena failed (" + std::to_string(want) + " bytes)");
            arena.cap = want;
        }
        Exec r; r.st = st; r.arena = &arena; r.dry = false; r.keep = debug; r.dbg = debug;
        if (debug) { trace.clear(); r.trace = &trace; }
        r.plan = &tail_plan;
        r.pool_base = arena.base + scratch_end;
        r.ticket_base = (unsigned*)(arena.base + scratch_end + pool_bytes);
        if (!fake && ticket_bytes && hipMemsetAsync(r.ticket_base, 0, ticket_bytes, st) != hipSuccess) return fail("hipMemsetAsync of the ticket pool failed");
        arena.off = 0; arena.peak = 0;
        r.prof = &prof; prof.used = 0;
        r.prof_gn = &prof_gn; prof_gn.used = 0; prof_gn.on = prof.on;
        fn(r);
        if (fake) fprintf(stderr, "[fake device] dry: tickets %zu pool %zu prod %d gn %d | real: tickets %zu pool %zu prod %d gn %d launches %lld\n", d.ticket_used,
                          d.pool_peak, d.prod_seq, d.gn_seq, r.ticket_used, r.pool_peak, r.prod_seq, r.gn_seq, r.launches);
        for (size_t k = 0; k < tail_plan.size(); ++k)
            if (tail_plan[k].on && !tail_plan[k].drawn) {
                if (fake) fprintf(stderr, "[fake device] planned tail of producer %zu (C %d, HW %d, consumer GroupNorm %d) was never attached\n", k, tail_plan[k].C, tail_plan[k].HW, tail_plan[k].consumer);
                r.err = -4; g_err = "internal: a planned GroupNorm tail was never attached to its producer";
            }
        if (r.ticket_used != d.ticket_used || r.pool_peak != d.pool_peak) { r.err = -4; g_err = "internal: dry and real pass disagree about the GroupNorm tail plan"; }
        last_launches = r.launches + (ticket_bytes ? 1 : 0);
        last_flops[0] = r.igemm_flops[0]; last_flops[1] = r.igemm_flops[1]; last_flops[2] = r.igemm_flops[2]; last_igemm_launches = r.igemm_launches;
        last_igemm_bytes = r.igemm_bytes;
        last_igemm_ms = 0.0;
        last_gn_ms = 0.0; last_gn_bytes = r.gn_bytes; last_gn_launches = r.gn_launches;
        for (int f = 0; f < Exec::F_COUNT; ++f) { last_fam[f][0] = r.fam_flops[f]; last_fam[f][1] = 0.0; last_fam[f][2] = (double)r.fam_launches[f]; }
        if (prof.on && prof.used) {
            // The event pair itself costs stream time (two marker packets bracket every launch): measure that cost with
            // empty pairs on the same stream and take it out, so that the per-launch figure is the kernel's own duration
            // (checked against the rocprofv3 kernel trace of the same command).
            constexpr int NCAL = 32;
            hipEvent_t cal[2 * NCAL];
            for (auto& ev : cal) (void)hipEventCreate(&ev);
            for (int i = 0; i < NCAL; ++i) { (void)hipEventRecord(cal[2 * i], st); (void)hipEventRecord(cal[2 * i + 1], st); }
            (void)hipStreamSynchronize(st);
            std::vector<float> empty(NCAL);
            for (int i = 0; i < NCAL; ++i) { empty[i] = 0.f; (void)hipEventElapsedTime(&empty[i], cal[2 * i], cal[2 * i + 1]); }
            for (auto& ev : cal) (void)hipEventDestroy(ev);
            std::sort(empty.begin(), empty.end());
            const float overhead = empty[NCAL / 2];   // median
            for (size_t i = 0; i + 1 < prof.used; i += 2) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, prof.ev[i], prof.ev[i + 1]);
                last_igemm_ms += std::max(0.f, ms - overhead);
                if (i / 2 < r.fam_of.size()) last_fam[r.fam_of[i / 2]][1] += std::max(0.f, ms - overhead);
            }
            if (!r.tag_of.empty()) {
                std::map<std::string, std::array<double, 3>> agg;
                for (size_t i = 0; i + 1 < prof.used && i / 2 < r.tag_of.size(); i += 2) {
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, prof.ev[i], prof.ev[i + 1]);
                    auto& a = agg[r.tag_of[i / 2]];
                    a[0] += std::max(0.f, ms - overhead); a[1] += 1.0; a[2] += r.tag_flops[i / 2];
                }
                std::vector<std::pair<double, std::string>> rows;
                last_shapes.clear();
                for (auto& kv : agg) {
                    char b[200]; snprintf(b, sizeof b, "%-40s n=%4.0f  %8.3f ms  %7.1f us/launch  %7.1f TF/s", kv.first.c_str(), kv.second[1], kv.second[0],
                                          1e3 * kv.second[0] / kv.second[1], kv.second[2] / (kv.second[0] * 1e-3) / 1e12);
                    rows.emplace_back(-kv.second[0], b);
                    char c[200]; snprintf(c, sizeof c, "shape %s n=%.0f ms=%.4f flops=%.6e\n", kv.first.c_str(), kv.second[1], kv.second[0], kv.second[2]);
                    last_shapes += c;
                }
                std::sort(rows.begin(), rows.end());
                static const bool shapes = getenv("RS_PROF_SHAPES") != nullptr;
                if (shapes) for (auto& rw : rows) fprintf(stderr, "[shapes] %s\n", rw.second.c_str());
            }
            // wall time of the parts (events at every change of part, the call's end closes the last one)
            if (!r.part_marks.empty()) {
                hipEvent_t endev = nullptr;
                (void)hipEventCreate(&endev); (void)hipEventRecord(endev, st); (void)hipEventSynchronize(endev);
                std::map<std::string, double> pm; std::vector<std::string> order;
                for (size_t i = 0; i < r.part_marks.size(); ++i) {
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, r.part_marks[i].second, i + 1 < r.part_marks.size() ? r.part_marks[i + 1].second : endev);
                    if (!pm.count(r.part_marks[i].first)) order.push_back(r.part_marks[i].first);
                    pm[r.part_marks[i].first] += ms;
                }
                for (auto& nm : order) { char c[96]; snprintf(c, sizeof c, "part %s ms=%.4f\n", nm.c_str(), pm[nm]); last_shapes += c; }
                (void)hipEventDestroy(endev);
            }
            for (size_t i = 0; i + 1 < prof_gn.used; i += 2) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, prof_gn.ev[i], prof_gn.ev[i + 1]);
                last_gn_ms += std::max(0.f, ms - overhead);
            }
        }
        for (auto& pmk : r.part_marks) (void)hipEventDestroy(pmk.second);
        if (r.err) return r.err;
        return 0;
    }
};

// ==================================================================== C ABI
extern "C" {

const char* rs_last_error(void) { return g_err.c_str(); }

rs_engine* rs_create(const rs_config* cfg) {
    if (!cfg) { g_err = "null config"; return nullptr; }
    const rs_unet_config& u = cfg->unet;
    if (!cfg->has_unet && !cfg->has_ae) { g_err = "config has neither a UNet nor an autoencoder"; return nullptr; }
    if (cfg->has_unet) {
        if (u.window_size != 8) { g_err = "only window_size 8 is supported"; return nullptr; }
        if (u.num_heads < 1 || u.swin_embed_dim != u.num_heads * 32) { g_err = "swin head dim must be 32"; return nullptr; }
        if (u.n_levels < 1 || u.n_levels > RS_MAX_LEVELS) { g_err = "bad n_levels"; return nullptr; }
        if ((u.image_size >> (u.n_levels - 1)) < 8) { g_err = "coarsest UNet level must be >= 8x8"; return nullptr; }
    }
    if (cfg->has_ae && cfg->ae.n_attn_res != 0) { g_err = "AE attn_resolutions must be empty"; return nullptr; }
    if (!cfg->enable_f16 && !cfg->enable_f32 && !cfg->enable_split) { g_err = "enable at least one precision"; return nullptr; }
    rs_engine* e = new rs_engine();
    e->cfg = *cfg;
    e->blob_bytes = e->build(nullptr, false);
    return e;
}

void rs_destroy(rs_engine* e) {
    if (!e) return;
    if (!rs_fake_device()) {
        for (auto& kv : e->film_cache) (void)hipFree(kv.second);
        if (e->arena.base) (void)hipFree(e->arena.base);
    }
    delete e;
}

int rs_load_tensor(rs_engine* e, const char* key, const float* host, const int64_t* shape, int ndim) {
    if (!e || !key || !host) return fail("rs_load_tensor: null argument");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host, host + n);
    e->host[key] = std::move(t);
    return 0;
}

size_t rs_weight_bytes(rs_engine* e) { return e ? e->blob_bytes : 0; }

int rs_bind_weight_blob(rs_engine* e, void* dev, size_t bytes) {
    if (!e || !dev) return fail("rs_bind_weight_blob: null argument");
    if (bytes < e->blob_bytes) return fail("weight blob too small");
    if (((uintptr_t)dev & 255) != 0) return fail("weight blob must be 256-byte aligned");
    e->build((char*)dev, false);  // resolve pointers
    e->collect_film_blocks();
    e->bound = true; e->ready = false;
    for (auto& kv : e->film_cache) (void)hipFree(kv.second);
    e->film_cache.clear();
    return 0;
}

// Broadcast the bound weight blob from rank `root` over an RCCL communicator the HOST owns (sampler.py:66-77: the reference's
// ranks each load the checkpoint - here rank 0 packs once and the blob travels over xGMI).  `rccl_comm` is an ncclComm_t; RCCL is looked up at
// call time (dlopen of librccl.so - the library has no link-time dependency on it: a single-GPU host never needs it).  One ncclBroadcast of
// rs_weight_bytes() bytes on `stream`, in place; the caller then calls rs_weights_ready() on every rank.  (The Python host mirror broadcasts the
// same buffer through torch.distributed, backend nccl = RCCL: resshift_amd/sharding.py.)
int rs_bcast_weights(rs_engine* e, void* rccl_comm, int root, void* stream) {
    if (!e || !e->bound || !rccl_comm) return fail("rs_bcast_weights: bind a weight blob first and pass an RCCL communicator");
    typedef int (*bcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static bcast_t fn = []() -> bcast_t {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        return h ? (bcast_t)dlsym(h, "ncclBroadcast") : nullptr;
    }();
    if (!fn) return fail("rs_bcast_weights: librccl.so (ncclBroadcast) not found");
    const int rc = fn(e->blob.base, e->blob.base, e->blob_bytes, /* ncclUint8 */ 1, root, rccl_comm, (hipStream_t)stream);
    if (rc != 0) return fail("rs_bcast_weights: ncclBroadcast failed (ncclResult_t " + std::to_string(rc) + ")");
    e->ready = false;   // (the receiving ranks' flags are read back by rs_weights_ready)
    return 0;
}

int rs_pack_weights(rs_engine* e) {
    if (!e || !e->bound) return fail("rs_pack_weights: bind a weight blob first");
    char* base = e->blob.base;
    e->build(base, true);
    e->collect_film_blocks();
    if (!e->build_err.empty()) { e->blob.staging.clear(); e->blob.staging.shrink_to_fit(); return fail(e->build_err); }
    e->split_ok = e->split_err.empty();
    {
        const uint32_t flags = e->split_ok ? 1u : 0u;
        memcpy(e->blob.staging.data(), &flags, sizeof flags);
    }
    if (hipMemcpy(base, e->blob.staging.data(), e->blob_bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy of weight blob failed");
    e->blob.staging.clear(); e->blob.staging.shrink_to_fit();
    e->blob.fill = false;
    e->host.clear();
    for (auto& kv : e->film_cache) (void)hipFree(kv.second);
    e->film_cache.clear();
    e->ready = true;
    return 0;
}

int rs_weights_ready(rs_engine* e) {
    if (!e || !e->bound) return fail("rs_weights_ready: bind a weight blob first");
    uint32_t flags = 0;   // header word of the blob (packed here, read from a cache file, or received by broadcast)
    if (rs_fake_device()) { e->split_ok = true; e->ready = true; return 0; }   // (plumbing check without a GPU, see run())
    if (hipMemcpy(&flags, e->blob.base, sizeof flags, hipMemcpyDeviceToHost) != hipSuccess) return fail("rs_weights_ready: cannot read the blob header");
    e->split_ok = (flags & 1u) != 0;
    if (e->conv_count > 0 && e->big_w_dev &&
        hipMemcpy(e->big_w.data(), e->big_w_dev, (size_t)e->conv_count, hipMemcpyDeviceToHost) != hipSuccess) return fail("rs_weights_ready: cannot read the per-layer flags");
    for (auto& kv : e->film_cache) (void)hipFree(kv.second);
    e->film_cache.clear();
    e->ready = true;
    return 0;
}

size_t rs_arena_bytes(rs_engine* e) { return e ? e->arena.cap : 0; }

// ---- profiling of the MFMA implicit-GEMM kernel family (bench.py roofline block)
int rs_profile_enable(rs_engine* e, int on) { if (!e) return -1; e->prof.on = on != 0; return 0; }
// out[0] = fp16-input igemm FLOPs of the last call, out[1] = fp32-input igemm FLOPs, out[2] = summed igemm kernel
// time in ms (hipEvents on the launch stream; 0 unless profiling was enabled), out[3] = igemm launch count,
// out[4] = algorithmic HBM bytes of those launches (each operand / result counted once), out[5] = split-input igemm FLOPs,
// out[6..8] = GroupNorm family: summed kernel ms (events as above), launch count (one per GroupNorm), algorithmic bytes
int rs_profile_get(rs_engine* e, double* out) {
    if (!e || !out) return -1;
    out[0] = e->last_flops[0]; out[1] = e->last_flops[1]; out[2] = e->last_igemm_ms; out[3] = (double)e->last_igemm_launches;
    out[4] = e->last_igemm_bytes; out[5] = e->last_flops[2];
    out[6] = e->last_gn_ms; out[7] = (double)e->last_gn_launches; out[8] = e->last_gn_bytes;
    return 0;
}

// per kernel family of the MFMA path, in this order: halo conv fp16 (igemm4), halo conv split, implicit GEMM fp16 (igemm2 / igemm3 /
// igemm), implicit GEMM split, implicit GEMM fp32, fused qkv + window attention + projection, fused Swin MLP, their split-storage
// variants, streaming AE attention:
// out[3 f + 0] = algorithmic FLOPs, out[3 f + 1] = kernel ms (0 unless profiling was on), out[3 f + 2] = launches.  Returns the family count.
int rs_profile_families(rs_engine* e, double* out, int cap) {
    if (!e || !out || cap < 3 * Exec::F_COUNT) return -1;
    for (int f = 0; f < Exec::F_COUNT; ++f) { out[3 * f] = e->last_fam[f][0]; out[3 * f + 1] = e->last_fam[f][1]; out[3 * f + 2] = e->last_fam[f][2]; }
    return Exec::F_COUNT;
}

// Text table of the last PROFILED call (rs_profile_enable): one line per distinct launch shape of the MFMA family,
//   "shape <part> f<family> M=.. N=.. K=.. z=.. n=<launches> ms=<summed kernel ms> flops=<algorithmic flops>"
// (part = encoder / unet / decoder; family numbering as rs_profile_families), then one line per part, "part <name> ms=<wall ms between the
// part's first launch and the next part's>" (every kernel of the part, whatever its family).  Returns the length needed (incl. the
// terminating 0); copies at most cap bytes.
int rs_profile_shapes(rs_engine* e, char* buf, int cap) {
    if (!e) return -1;
    const int need = (int)e->last_shapes.size() + 1;
    if (buf && cap > 0) { const int n = std::min(cap - 1, need - 1); memcpy(buf, e->last_shapes.data(), n); buf[n] = 0; }
    return need;
}

// ---- debug trace (tests only): record named intermediate activations of the next network call
int rs_debug_enable(rs_engine* e, int on) { if (!e) return -1; e->debug = on != 0; e->trace.clear(); return 0; }
int rs_debug_count(rs_engine* e) { return e ? (int)e->trace.size() : 0; }
int rs_debug_info(rs_engine* e, int i, char* name, int cap, int* dims /* B,C,H,W */) {
    if (!e || i < 0 || i >= (int)e->trace.size()) return -1;
    const auto& t = e->trace[i];
    snprintf(name, cap, "%s", t.first.c_str());
    dims[0] = t.second.B; dims[1] = t.second.C; dims[2] = t.second.H; dims[3] = t.second.W;
    return 0;
}
int rs_debug_fetch(rs_engine* e, int i, float* out_nchw, void* stream) {
    if (!e || i < 0 || i >= (int)e->trace.size()) return -1;
    const View& v = e->trace[i].second;
    return rs_nhwc_to_nchw_launch(v.p, v.dt, out_nchw, v.B, v.C, v.H * v.W, v.ld, 0, (hipStream_t)stream);
}
long long rs_last_launch_count(rs_engine* e) { return e ? e->last_launches : 0; }

int rs_unet_forward(rs_engine* e, const float* x, const int* t_host, const float* lq, const float* mask, float* out, int B, int H, int W,
                    int Hl, int Wl, int prec, void* stream) {
    if (!e || !e->cfg.has_unet) return fail("engine has no UNet");
    {
        const int sh = e->cfg.unet.n_levels - 1;
        if ((H % (8 << sh)) || (W % (8 << sh))) return fail("UNet input H/W must be multiples of 8*2^(levels-1)");
    }
    hipStream_t st = (hipStream_t)stream;
    if (!e->ready) return fail("weights are not ready");
    if (B < 1 || !x || !out || !t_host) return fail("rs_unet_forward: bad batch / null tensor");
    if (e->cfg.unet.cond_lq && !lq) return fail("rs_unet_forward: this UNet is conditioned on lq (cond_lq) but lq is NULL");
    if (e->cfg.unet.cond_mask && !mask) return fail("rs_unet_forward: this UNet is conditioned on a mask (cond_mask) but mask is NULL");
    if (e->cfg.unet.cond_lq && e->fe_convs.empty() && (Hl != H || Wl != W))
        return fail("rs_unet_forward: without a feature extractor lq must have the latent resolution");
    if (!e->fe_convs.empty() && ((Hl >> (int)e->fe_convs.size()) != H || (Wl >> (int)e->fe_convs.size()) != W))
        return fail("rs_unet_forward: lq resolution does not match the feature extractor's down-sampling");
    for (int b = 1; b < B; ++b) if (t_host[b] != t_host[0]) return fail("rs_unet_forward: per-sample timesteps must be equal within a batch");
    const float* film = e->film_row(t_host[0], st);
    if (!film) return fail("FiLM table allocation failed");
    return e->run(st, [&](Exec& ex) {
        View feat; const View* fp = nullptr;
        if (!e->fe_convs.empty()) { feat = e->feature_extract(ex, lq, mask, B, Hl, Wl, prec); fp = &feat; }
        e->unet_body(ex, x, 1.0f, fp, lq, mask, Hl, Wl, out, B, H, W, prec, film);
    });
}

int rs_vq_encode(rs_engine* e, const float* img, float* z, int B, int H, int W, int prec, void* stream) {
    if (!e || !e->cfg.has_ae) return fail("engine has no autoencoder");
    hipStream_t st = (hipStream_t)stream;
    return e->run(st, [&](Exec& ex) {
        View in = ex.T(B, H, W, e->enc_in.CinP, prec);  // RGB zero-padded to 8 channels in the working precision
        e->zero(ex, in);
        if (!ex.dry) ex.check(rs_nchw_to_nhwc_launch(img, in.p, prec, B, e->cfg.ae.in_channels, H * W, in.ld, 0, 1.f, st), "img->nhwc");
        e->encode_body(ex, in, z, prec);
    });
}

int rs_vq_decode(rs_engine* e, const float* z, float* img, int32_t* idx_out, int B, int h, int w, int force_not_quantize, int prec,
                 void* stream) {
    if (!e || !e->cfg.has_ae) return fail("engine has no autoencoder");
    hipStream_t st = (hipStream_t)stream;
    return e->run(st, [&](Exec& ex) { e->decode_body(ex, z, 1.0f, img, idx_out, B, h, w, force_not_quantize, prec); });
}

int rs_bicubic(rs_engine* e, const float* y, float* out, int B, int C, int H, int W, int sf, void* stream) {
    if (!e) return fail("null engine");
    hipStream_t st = (hipStream_t)stream;
    const bool was_ready = e->ready;
    e->ready = true;  // no weights involved
    const int rc = e->run(st, [&](Exec& ex) {
        View t = ex.T(B, H * sf, W * sf, C, RS_F32);
        if (ex.dry) return;
        ex.check(rs_bicubic_launch(y, t.p, RS_F32, B, C, H, W, sf, t.ld, st), "bicubic");
        ex.check(rs_nhwc_to_nchw_launch(t.p, RS_F32, out, B, C, H * sf * W * sf, t.ld, 0, st), "bicubic->nchw");
    });
    e->ready = was_ready;
    return rc;
}

int rs_axpbypcz(const float* x, const float* z, const float* n, float* y, float a, float b, float c, long long count, void* stream) {
    return rs_axpbypcz_launch(x, z, n, y, a, b, c, count, (hipStream_t)stream);
}

// gaussian_diffusion.py:367-472: encode_first_stage(up_sample) -> prior_sample -> T x p_sample -> decode_first_stage
int rs_sample(rs_engine* e, const rs_sample_args* a) {
    if (!e || !a) return fail("null argument");
    if (!e->cfg.has_ae || !e->cfg.has_unet) return fail("rs_sample needs both the UNet and the autoencoder");
    if (a->steps < 1 || a->steps > RS_MAX_STEPS) return fail("bad step count");
    {
        auto bad = [](int p) { return p != RS_F16 && p != RS_F32 && p != RS_F16S; };
        if (bad(a->prec_encode) || bad(a->prec_decode)) return fail("bad precision");
        for (int i = 0; i < a->steps; ++i) if (bad(a->prec_unet[i])) return fail("bad precision");
    }
    hipStream_t st = (hipStream_t)a->stream;
    if (!e->ready) return fail("weights are not ready");
    const rs_ae_config& ae = e->cfg.ae;
    const int f = 1 << (ae.n_levels - 1);
    const int B = a->B, Hi = a->h * a->sf, Wi = a->w * a->sf, hz = Hi / f, wz = Wi / f, Cz = ae.embed_dim;
    if (e->cfg.unet.in_channels != Cz) return fail("UNet in_channels != AE embed_dim");
    if (B < 1 || a->h < 1 || a->w < 1 || a->sf < 1 || !a->y || !a->noise || !a->out) return fail("rs_sample: bad batch / size / null tensor");
    if ((Hi % f) || (Wi % f)) return fail("rs_sample: h*sf and w*sf must be multiples of the autoencoder's down-sampling factor");
    {
        const int sh = e->cfg.unet.n_levels - 1;
        if ((hz % (8 << sh)) || (wz % (8 << sh))) return fail("rs_sample: latent H/W must be multiples of 8*2^(levels-1) (pad the input, sampler.py:130-138)");
    }
    if (e->cfg.unet.cond_lq && e->fe_convs.empty() && (a->h != hz || a->w != wz))
        return fail("rs_sample: without a feature extractor the LQ image is concatenated at latent resolution: h*sf/f must equal h");
    if (e->cfg.unet.cond_mask && !a->mask) return fail("rs_sample: this UNet is conditioned on a mask (cond_mask) but mask is NULL");
    std::vector<const float*> films(a->steps);
    for (int t = 0; t < a->steps; ++t) {
        films[t] = e->film_row(a->tmap[t], st);
        if (!films[t]) return fail("FiLM table allocation failed");
    }
    const long long zcount = (long long)B * Cz * hz * wz;
    return e->run(st, [&](Exec& ex) {
        float* z_y = (float*)ex.raw(zcount * 4);
        float* xt = (float*)ex.raw(zcount * 4);
        float* pred = (float*)ex.raw(zcount * 4);
        // conditioning for the UNet: raw lq at latent resolution, or the feature-extractor output (step invariant: hoisted)
        View feat[3]; bool have_feat[3] = {false, false, false};
        if (!e->fe_convs.empty())
            for (int i = 0; i < a->steps; ++i) {
                const int pr = a->prec_unet[i];
                if (!have_feat[pr]) { feat[pr] = e->feature_extract(ex, a->y, a->mask, B, a->h, a->w, pr); have_feat[pr] = true; }
            }
        // encode_first_stage (gaussian_diffusion.py:500-515)
        {
            const size_t mk = ex.mark();
            View in = ex.T(B, Hi, Wi, e->enc_in.CinP, a->prec_encode);  // RGB zero-padded to 8 channels
            e->zero(ex, in);
            if (!ex.dry) {
                if (a->sf != 1) ex.check(rs_bicubic_launch(a->y, in.p, a->prec_encode, B, ae.in_channels, a->h, a->w, a->sf, in.ld, st), "bicubic");
                else ex.check(rs_nchw_to_nhwc_launch(a->y, in.p, a->prec_encode, B, ae.in_channels, Hi * Wi, in.ld, 0, 1.f, st), "y->nhwc");
            }
            e->encode_body(ex, in, z_y, a->prec_encode);
            ex.reset(mk);
        }
        if (!ex.dry) {
            // z_y * scale_factor, then prior_sample: x_T = z_y + kappa*sqrt(eta_T)*noise (gaussian_diffusion.py:512,529)
            if (a->scale_factor != 1.0f) ex.check(rs_axpbypcz_launch(z_y, nullptr, nullptr, z_y, a->scale_factor, 0.f, 0.f, zcount, st), "scale z_y");
            ex.check(rs_axpbypcz_launch(z_y, nullptr, a->noise, xt, 1.f, 0.f, a->prior_scale, zcount, st), "prior_sample");
        }
        for (int i = a->steps - 1, k = 1; i >= 0; --i, ++k) {
            // model(_scale_input(x_t, t), t, lq) -> pred_xstart (gaussian_diffusion.py:266,278)
            const int pr = a->prec_unet[i];
            e->unet_body(ex, xt, a->inv_std[i], e->fe_convs.empty() ? nullptr : &feat[pr], a->y, a->mask, a->h, a->w, pred, B, hz, wz, pr, films[i]);
            if (!ex.dry) {
                // mean = c1*x_t + c2*x0 (:218-221); sample = mean + [t>0]*sigma_t*eps (:358-364)
                const float* nz = (i > 0) ? a->noise + (long long)k * zcount : nullptr;
                ex.check(rs_axpbypcz_launch(xt, pred, nz, xt, a->coef1[i], a->coef2[i], a->sigma[i], zcount, st), "posterior step");
            }
        }
        if (a->z_out && !ex.dry) (void)hipMemcpyAsync(a->z_out, xt, zcount * 4, hipMemcpyDeviceToDevice, st);
        // decode_first_stage: z / scale_factor -> VQ -> post_quant_conv -> Decoder (gaussian_diffusion.py:474-498)
        e->decode_body(ex, xt, 1.0f / a->scale_factor, a->out, a->idx_out, B, hz, wz, 0, a->prec_decode);
    });
}

// -------------------------------------------------------------------- op-level test entry points
static void* dev_copy(const void* host, size_t bytes) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
    return d;
}

int rs_op_conv2d(const void* x0, const void* x1, const float* w_ref_host, const float* bias_host, const void* res, void* y, int B, int Hs,
                 int Ws, int C0, int C1, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int up, int act,
                 int in_prec, int out_prec, int force_direct, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Cin = C0 + C1;
    const size_t K = (size_t)KH * KW * Cin, n = K * Cout;
    const bool direct = force_direct || (Cin % 8 != 0) || (C0 % 8 != 0) || Cout <= 8;
    float* bias = bias_host ? (float*)dev_copy(bias_host, Cout * 4) : nullptr;
    int rc;
    void* wdev = nullptr;
    if (direct) {
        std::vector<float> o(n);
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                for (int t = 0; t < KH * KW; ++t) o[((size_t)t * Cin + ci) * Cout + co] = w_ref_host[((size_t)co * Cin + ci) * KH * KW + t];
        wdev = dev_copy(o.data(), n * 4);
        DirectConvParams p{};
        p.x0 = x0; p.x1 = x1; p.w = (const float*)wdev; p.bias = bias; p.y = y; p.C0 = C0; p.C1 = C1; p.ld0 = C0; p.ld1 = C1;
        p.B = B; p.Hs = Hs; p.Ws = Ws; p.up = up; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
        p.Cout = Cout; p.ldy = Cout; p.act = act;
        if (res) { rc = fail("direct conv has no residual path"); }
        else rc = rs_direct_conv_launch(&p, in_prec, out_prec, st);
    } else {
        if (in_prec == RS_F16) {
            std::vector<f16> o(n);
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int t = 0; t < KH * KW; ++t) o[(size_t)co * K + (size_t)t * Cin + ci] = (f16)w_ref_host[((size_t)co * Cin + ci) * KH * KW + t];
            wdev = dev_copy(o.data(), n * 2);
        } else if (in_prec == RS_F16S) {
            std::vector<f16> o(2 * n);   // [Cout][K hi | K lo]
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int t = 0; t < KH * KW; ++t) {
                        f16 h, l;
                        rs_split(w_ref_host[((size_t)co * Cin + ci) * KH * KW + t], h, l);
                        o[(size_t)co * 2 * K + (size_t)t * Cin + ci] = h;
                        o[(size_t)co * 2 * K + K + (size_t)t * Cin + ci] = l;
                    }
            wdev = dev_copy(o.data(), n * 4);
        } else {
            std::vector<float> o(n);
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int t = 0; t < KH * KW; ++t) o[(size_t)co * K + (size_t)t * Cin + ci] = w_ref_host[((size_t)co * Cin + ci) * KH * KW + t];
            wdev = dev_copy(o.data(), n * 4);
        }
        IGemmParams p{};
        p.x0 = x0; p.x1 = x1; p.w = wdev; p.bias = bias; p.res = res; p.y = y; p.C0 = C0; p.C1 = C1; p.ld0 = C0; p.ld1 = C1;
        p.B = B; p.Hs = Hs; p.Ws = Ws; p.up = up; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
        p.Cout = Cout; p.ldy = Cout; p.ldres = Cout; p.M = B * Ho * Wo; p.Ktot = (int)K; p.act = act; p.out_scale = 1.f;
        {
            int tw4, bc4, seg4, sk4;   // the halo kernel plans its own split-K (as Engine::conv does)
            if (rs_igemm4_plan(&p, in_prec, out_prec, 1, &tw4, &bc4, &seg4, &sk4)) p.splitk = sk4;
            else p.splitk = rs_igemm_splitk_plan(p.M, Cout, (int)K, in_prec);
        }
        float* part = nullptr;
        if (p.splitk > 1) { (void)hipMalloc((void**)&part, (size_t)p.splitk * p.M * Cout * sizeof(float)); p.partial = part; }
        rc = rs_igemm_launch(&p, in_prec, out_prec, 1, st);
        if (rc) fail("igemm launch rejected the shape");
        (void)hipStreamSynchronize(st);
        if (part) (void)hipFree(part);
    }
    (void)hipStreamSynchronize(st);
    if (wdev) (void)hipFree(wdev);
    if (bias) (void)hipFree(bias);
    return rc;
}

// micro-benchmark of one implicit-GEMM conv shape (random device data is supplied by the caller): `reps` launches
// bracketed by hipEvents on the stream; returns the average milliseconds per launch in *ms_out.
int rs_op_conv2d_bench(const void* x0, const void* w_packed_dev, const float* bias_dev, const void* res, void* y, int B, int Hs, int Ws,
                       int Cin, int Cout, int KH, int KW, int stride, int pad, int Ho, int Wo, int up, int act, int in_prec, int out_prec,
                       int reps, float* ms_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    IGemmParams p{};
    p.x0 = x0; p.w = w_packed_dev; p.bias = bias_dev; p.res = res; p.y = y; p.C0 = Cin; p.ld0 = Cin;
    p.B = B; p.Hs = Hs; p.Ws = Ws; p.up = up; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad_t = pad; p.pad_l = pad;
    p.Cout = Cout; p.ldy = Cout; p.ldres = Cout; p.M = B * Ho * Wo; p.Ktot = KH * KW * Cin; p.act = act; p.out_scale = 1.f;
    {
        int tw4, bc4, seg4, sk4;
        if (rs_igemm4_plan(&p, in_prec, out_prec, 1, &tw4, &bc4, &seg4, &sk4)) p.splitk = sk4;
        else p.splitk = rs_igemm_splitk_plan(p.M, Cout, p.Ktot, in_prec);
    }
    float* part = nullptr;
    if (p.splitk > 1) { (void)hipMalloc((void**)&part, (size_t)p.splitk * p.M * Cout * sizeof(float)); p.partial = part; }
    int rc = rs_igemm_launch(&p, in_prec, out_prec, 1, st);  // warm-up
    if (rc) return fail("igemm launch rejected the shape");
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) rc |= rs_igemm_launch(&p, in_prec, out_prec, 1, st);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms / (float)std::max(1, reps);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (part) (void)hipFree(part);
    return rc;
}

// GroupNorm-affine + SiLU + 3x3 conv on the halo kernel (igemm4.hip): x raw fp16 NHWC, coef_dev [B][2][Cin] fp32 (scale row,
// shift row) or null, weights in the reference layout on the host; fails when the shape is not eligible for that kernel
int rs_op_conv3x3_halo(const void* x, const float* coef_dev, int act_in, const float* w_ref_host, const float* bias_host, const void* res, void* y,
                       int B, int H, int W, int Cin, int Cout, int prec, float* ystats_dev, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t K = (size_t)9 * Cin, n = K * Cout;
    if (prec != RS_F16 && prec != RS_F16S) return fail("halo kernel: fp16 or split storage");
    const bool sp = prec == RS_F16S;
    std::vector<f16> o(sp ? 2 * n : n);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const float wv = w_ref_host[((size_t)co * Cin + ci) * 9 + t];
                if (!sp) { o[(size_t)co * K + (size_t)t * Cin + ci] = (f16)wv; continue; }
                f16 h, l;
                rs_split(wv, h, l);
                o[(size_t)co * 2 * K + (size_t)t * Cin + ci] = h;
                o[(size_t)co * 2 * K + K + (size_t)t * Cin + ci] = l;
            }
    void* wdev = dev_copy(o.data(), o.size() * 2);
    float* bias = bias_host ? (float*)dev_copy(bias_host, Cout * 4) : nullptr;
    IGemmParams p{};
    p.x0 = x; p.w = wdev; p.bias = bias; p.res = res; p.y = y; p.C0 = Cin; p.ld0 = Cin; p.B = B; p.Hs = H; p.Ws = W; p.up = 1; p.Ho = H; p.Wo = W;
    p.KH = 3; p.KW = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1; p.Cout = Cout; p.ldy = Cout; p.ldres = Cout; p.M = B * H * W; p.Ktot = (int)K;
    p.out_scale = 1.f; p.splitk = 1; p.xcoef = coef_dev; p.xact = act_in;
    int tw, bc, seg, sk, rc;
    float* part = nullptr;
    if (!rs_igemm4_plan(&p, prec, prec, 1, &tw, &bc, &seg, &sk)) rc = fail("shape is not eligible for the halo kernel");
    else {
        p.splitk = sk;
        if (sk > 1) { (void)hipMalloc((void**)&part, (size_t)sk * p.M * Cout * sizeof(float)); p.partial = part; }
        p.ystats = ystats_dev; p.ystats_ld = Cout;
        rc = rs_igemm_launch(&p, prec, prec, 1, st);
    }
    (void)hipStreamSynchronize(st);
    if (part) (void)hipFree(part);
    if (wdev) (void)hipFree(wdev);
    if (bias) (void)hipFree(bias);
    return rc;
}

// The same layer on the Winograd F(2x2,3x3) kernel (wino.hip; split storage only): weights transformed and packed on the host, one checked
// launch; with reps > 0 the launch is then repeated `reps` times between two hipEvents and *ms_out receives the average milliseconds.
// `ystats_dev`: [B][H*W / 128][Cout][2] (one slab per 8 x 16 pixel tile).  Fails when the shape is not eligible.
int rs_op_conv3x3_wino(const void* x, const float* coef_dev, int act_in, const float* w_ref_host, const float* bias_host, const void* res, void* y,
                       int B, int H, int W, int Cin, int Cout, float* ystats_dev, int reps, float* ms_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (Cin < 32 || (Cin % 32) || Cout < 32 || (Cout % 32)) return fail("wino kernel: channels in multiples of 32");
    std::vector<char> packed(rs_wino_weight_bytes(Cin, Cout));
    const float mx = rs_wino_pack(w_ref_host, Cin, Cout, packed.data());
    if (!(mx < 30.0f)) return fail("wino kernel: |U| >= 30");
    void* wdev = dev_copy(packed.data(), packed.size());
    float* bias = bias_host ? (float*)dev_copy(bias_host, Cout * 4) : nullptr;
    IGemmParams p{};
    p.x0 = x; p.ww = wdev; p.bias = bias; p.res = res; p.y = y; p.C0 = Cin; p.ld0 = Cin; p.B = B; p.Hs = H; p.Ws = W; p.up = 1; p.Ho = H; p.Wo = W;
    p.KH = 3; p.KW = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1; p.Cout = Cout; p.ldy = Cout; p.ldres = Cout; p.M = B * H * W; p.Ktot = 9 * Cin;
    p.out_scale = 1.f; p.splitk = 1; p.xcoef = coef_dev; p.xact = act_in; p.ystats = ystats_dev; p.ystats_ld = Cout;
    int rc = 0;
    float* stamps = nullptr;   // RS_WINO_STAMPS=1 with a -DRS_WINO_PHASES build: per-workgroup phase cycles of wave 0 (see wino.hip), averaged to stderr
    const int ntile = rs_wino_tiles(&p);
    p.dbg = 64;   // (an op-level entry: no fill-the-chip threshold)
    if (const char* ab = getenv("RS_WINO_ABL")) p.dbg |= atoi(ab);   // (-DRS_WINO_PHASES builds: timing ablations, wino.hip)
    if (getenv("RS_WINO_STAMPS")) { (void)hipMalloc((void**)&stamps, (size_t)ntile * 16 * sizeof(float)); (void)hipMemset(stamps, 0, (size_t)ntile * 16 * sizeof(float)); p.partial = stamps; }
    if (!rs_wino_plan(&p, RS_F16S, RS_F16S, 1)) rc = fail("shape is not eligible for the wino kernel");
    else {
        rc = rs_wino_launch(&p, st);
        if (rc) fail("wino launch failed");
        if (stamps) {
            (void)hipStreamSynchronize(st);
            std::vector<float> hs((size_t)ntile * 16);
            (void)hipMemcpy(hs.data(), stamps, hs.size() * sizeof(float), hipMemcpyDeviceToHost);
            double wide[16] = {}; int nw = 0;
            for (int t = 0; t < ntile; ++t) {
                for (int i = 0; i < 16; ++i) wide[i] += hs[(size_t)t * 16 + i];
                ++nw;
            }
            static const char* nm[8] = {"prologue", "wait+barrier", "halo issue", "B operand", "MFMA steps", "conversion", "epilogue", "total"};
            fprintf(stderr, "[wino phases] %dx%dx%d %d->%d, mean cycles of wave 0 over %d workgroups:", B, H, W, Cin, Cout, nw);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f", nm[i], wide[i] / std::max(1, nw));
            static const char* nq[6] = {"drain", "barrier-1", "exchange writes", "barrier-2", "transform + stores", "statistics"};
            fprintf(stderr, "  | epilogue:");
            for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.0f", nq[i], wide[8 + i] / std::max(1, nw));
            fprintf(stderr, "\n");
        }
        if (!rc && reps > 0) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < reps; ++i) rc |= rs_wino_launch(&p, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms_out) *ms_out = ms / (float)reps;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
    (void)hipStreamSynchronize(st);
    if (stamps) (void)hipFree(stamps);
    if (wdev) (void)hipFree(wdev);
    if (bias) (void)hipFree(bias);
    return rc;
}

// pixels per statistics slab that rs_op_conv3x3_halo would use for this shape (0: not eligible / no statistics): the caller sizes ystats_dev
// as [B][H*W / slab][Cout][2]
int rs_op_conv3x3_halo_stats_px(int B, int H, int W, int Cin, int Cout, int prec) {
    IGemmParams p{};
    p.C0 = Cin; p.ld0 = Cin; p.B = B; p.Hs = H; p.Ws = W; p.up = 1; p.Ho = H; p.Wo = W;
    p.KH = 3; p.KW = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1; p.Cout = Cout; p.ldy = Cout; p.ldres = Cout; p.M = B * H * W; p.Ktot = 9 * Cin;
    return rs_igemm4_stats_px(&p, prec);
}

int rs_op_gemm_nt(const void* a, const void* b, const float* bias_dev, void* y, int nz, int M, int N, int K, float scale, int in_prec,
                  int out_prec, void* stream) {
    IGemmParams p{};
    p.x0 = a; p.w = b; p.bias = bias_dev; p.y = y; p.C0 = K; p.ld0 = K; p.B = 1; p.Hs = M; p.Ws = 1; p.up = 1; p.Ho = M; p.Wo = 1;
    p.KH = 1; p.KW = 1; p.stride = 1; p.Cout = N; p.ldy = N; p.M = M; p.Ktot = K; p.out_scale = scale;
    p.bs_x0 = (long long)M * K; p.bs_w = (long long)N * K; p.bs_y = (long long)M * N;
    const int rc = rs_igemm_launch(&p, in_prec, out_prec, nz, (hipStream_t)stream);
    if (rc) fail("igemm launch rejected the shape");
    return rc;
}

int rs_op_groupnorm(const void* x, void* y, const float* gamma_host, const float* beta_host, const float* film_dev, int B, int HW, int C,
                    int groups, float eps, int act, int prec, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float* g = (float*)dev_copy(gamma_host, C * 4);
    float* bt = (float*)dev_copy(beta_host, C * 4);
    int S = std::max(1, std::min(64, 1024 / std::max(1, B)));
    S = std::max(1, std::min(S, HW / 8));
    const int S2 = std::max(1, std::min(HW / 8, std::max(1, 2048 / std::max(1, B))));
    float* partial = nullptr;
    (void)hipMalloc((void**)&partial, (size_t)B * S * groups * 2 * 4);
    GNParams p{};
    p.x = x; p.y = y; p.gamma = g; p.beta = bt; p.film = film_dev; p.partial = partial; p.B = B; p.HW = HW; p.C = C; p.ldx = C; p.ldy = C;
    p.S = S; p.groups = groups; p.eps = eps; p.act = act;
    const int nk = rs_groupnorm_launch(&p, prec, S2, st);   // kernels launched, or < 0
    const int rc = nk < 0 ? nk : 0;
    if (rc) fail("groupnorm launch rejected the shape");
    (void)hipStreamSynchronize(st);
    (void)hipFree(g); (void)hipFree(bt); (void)hipFree(partial);
    return rc;
}

int rs_op_window_attention(const void* qkv, void* out, const float* table_host, int B, int H, int W, int heads, int shift, int prec,
                           void* stream) {
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> bt((size_t)heads * 64 * 64);
    for (int h = 0; h < heads; ++h)
        for (int j = 0; j < 64; ++j)
            for (int i = 0; i < 64; ++i) {
                const int idx = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
                bt[((size_t)h * 64 + j) * 64 + i] = table_host[(size_t)idx * heads + h];
            }
    float* d = (float*)dev_copy(bt.data(), bt.size() * 4);
    std::vector<float> bn(bt.size());
    for (int h = 0; h < heads; ++h)
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) bn[((size_t)h * 64 + i) * 64 + j] = bt[((size_t)h * 64 + j) * 64 + i];
    float* dn = (float*)dev_copy(bn.data(), bn.size() * 4);
    WinAttnParams p{};
    p.bias_n = dn;
    p.qkv = qkv; p.out = out; p.bias_t = d; p.B = B; p.H = H; p.W = W; p.heads = heads; p.shift = shift; p.ldq = 3 * heads * 32;
    p.ldo = heads * 32; p.scale = 1.0f / std::sqrt(32.0f);
    const int rc = rs_win_attn_launch(&p, prec, st);
    if (rc) fail("window attention launch rejected the shape");
    (void)hipStreamSynchronize(st);
    (void)hipFree(d);
    (void)hipFree(dn);
    return rc;
}

int rs_op_window_attention_qkv(const void* x, const void* wqkv_dev, const float* bqkv_dev, const void* wproj_dev, const float* bproj_dev,
                               const void* res, void* out, const float* table_host, int B, int H, int W, int heads, int shift, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> bn((size_t)heads * 64 * 64);
    for (int h = 0; h < heads; ++h)
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                const int idx = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
                bn[((size_t)h * 64 + i) * 64 + j] = table_host[(size_t)idx * heads + h];
            }
    float* dn = (float*)dev_copy(bn.data(), bn.size() * 4);
    std::vector<float> bc((size_t)heads * 256, 0.0f);   // WinAttnParams::bias_c
    for (int h = 0; h < heads; ++h)
        for (int k = 0; k < 225; ++k) bc[(size_t)h * 256 + k] = table_host[(size_t)k * heads + h] * 1.44269504088896f;
    float* dc = (float*)dev_copy(bc.data(), bc.size() * 4);
    WinAttnParams p{};
    p.bias_n = dn; p.bias_c = dc; p.out = out; p.B = B; p.H = H; p.W = W; p.heads = heads; p.shift = shift; p.ldo = heads * 32; p.scale = 1.0f / std::sqrt(32.0f);
    // the kernel reads its weights in fragment-major order (ConvW::wh_frag): repack the caller's row-major operands
    const int E = heads * 32;
    void* wq_f = frag_major_from_device_rows(wqkv_dev, 3 * E, E, E, -1);
    void* wp_f = wproj_dev ? frag_major_from_device_rows(wproj_dev, E, E, E, -1) : nullptr;
    p.x = x; p.wqkv = wq_f; p.bqkv = bqkv_dev; p.ldx = heads * 32;
    p.wproj = wp_f; p.bproj = bproj_dev; p.res = res; p.ldres = heads * 32;
    const int rc = (wq_f && (wp_f || !wproj_dev)) ? rs_win_attn_qkv_launch(&p, st) : -1;
    if (rc) fail("fused qkv + window attention launch rejected the shape (fp16, 6 heads of 32 only)");
    (void)hipStreamSynchronize(st);
    (void)hipFree(dn); (void)hipFree(dc); (void)hipFree(wq_f); (void)hipFree(wp_f);
    return rc;
}

int rs_op_window_attention_qkv_split(const void* x, const void* wqkv_dev, const float* bqkv_dev, const void* wproj_dev, const float* bproj_dev,
                                     const void* res, void* out, const float* table_host, const float* xcoef_dev, int B, int H, int W, int heads,
                                     int shift, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> bn((size_t)heads * 64 * 64);
    for (int h = 0; h < heads; ++h)
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                const int idx = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
                bn[((size_t)h * 64 + i) * 64 + j] = table_host[(size_t)idx * heads + h];
            }
    float* dn = (float*)dev_copy(bn.data(), bn.size() * 4);
    std::vector<float> bc((size_t)heads * 256, 0.0f);   // WinAttnParams::bias_c
    for (int h = 0; h < heads; ++h)
        for (int k = 0; k < 225; ++k) bc[(size_t)h * 256 + k] = table_host[(size_t)k * heads + h] * 1.44269504088896f;
    float* dc = (float*)dev_copy(bc.data(), bc.size() * 4);
    WinAttnParams p{};
    p.bias_n = dn; p.bias_c = dc; p.out = out; p.B = B; p.H = H; p.W = W; p.heads = heads; p.shift = shift; p.ldo = heads * 32; p.scale = 1.0f / std::sqrt(32.0f);
    // fragment-major (hi, lo) weights (ConvW::ws_frag) from the caller's rows [K hi | K lo]
    const int E = heads * 32;
    void* wq_f = frag_major_from_device_rows(wqkv_dev, 3 * E, E, 2 * E, E);
    void* wp_f = wproj_dev ? frag_major_from_device_rows(wproj_dev, E, E, 2 * E, E) : nullptr;
    p.x = x; p.wqkv = wq_f; p.bqkv = bqkv_dev; p.ldx = heads * 32; p.xcoef = xcoef_dev;
    p.wproj = wp_f; p.bproj = bproj_dev; p.res = res; p.ldres = heads * 32;
    const int rc = (wq_f && (wp_f || !wproj_dev)) ? rs_win_attn_qkv_split_launch(&p, st) : -1;
    if (rc) fail("fused split qkv + window attention launch rejected the shape (split storage, 6 heads of 32 only)");
    (void)hipStreamSynchronize(st);
    (void)hipFree(dn); (void)hipFree(dc); (void)hipFree(wq_f); (void)hipFree(wp_f);
    return rc;
}

int rs_op_ae_flash_attention(const void* q, const void* k, const void* vt, const float* bv_dev, void* o, int nz, int T, int C, void* stream) {
    const int rc = rs_ae_flash_launch(q, C, k, C, vt, bv_dev, o, C, nz, T, C, 1.0f / std::sqrt((float)C), (hipStream_t)stream);
    if (rc) fail("streaming AE attention launch rejected the shape (fp16, C in {128, 256, 512}, T a multiple of 128)");
    return rc;
}

int rs_op_ae_flash_attention_split(const void* q, const void* k, const void* vt, const float* bv_dev, void* o, int nz, int T, int C, void* stream) {
    const int rc = rs_ae_flash_split_launch(q, C, k, C, vt, bv_dev, o, C, nz, T, C, 1.0f / std::sqrt((float)C), (hipStream_t)stream);
    if (rc) fail("split-storage streaming AE attention launch rejected the shape (C = 512, T a multiple of 64)");
    return rc;
}

int rs_op_swin_mlp(const void* x, const void* w1_dev, const float* b1_dev, const void* w2_dev, const float* b2_dev, const void* res, void* y,
                   int M, int E, int HD, void* stream) {
    const int rc = rs_swin_mlp_launch(x, w1_dev, b1_dev, w2_dev, b2_dev, res, y, M, E, E, E, E, HD, nullptr, 0, nullptr, 0, (hipStream_t)stream);
    if (rc) fail("swin_mlp launch rejected the shape (fp16, E = 192, HD = 768 only)");
    return rc;
}
int rs_op_swin_mlp_split(const void* x, const void* w1_dev, const float* b1_dev, const void* w2_dev, const float* b2_dev, const void* res, void* y,
                         int M, int E, int HD, void* stream) {
    const int rc = rs_swin_mlp_split_launch(x, w1_dev, b1_dev, w2_dev, b2_dev, res, y, M, E, E, E, E, HD, nullptr, 0, nullptr, 0, nullptr, (hipStream_t)stream);
    if (rc) fail("swin_mlp_split launch rejected the shape (split storage, E = 192, HD = 768 only)");
    return rc;
}
int rs_op_swin_mlp_split_unembed(const void* x, const float* xcoef_dev, const void* w1_dev, const float* b1_dev, const void* w2cat_dev, const float* bcat_dev,
                                 void* y, int M, int HW, int E, int HD, int NO, void* stream) {
    const int rc = rs_swin_mlp_split_launch_n(x, w1_dev, b1_dev, w2cat_dev, bcat_dev, nullptr, y, M, E, 0, NO, E, HD, NO, xcoef_dev, HW, nullptr, 0, nullptr,
                                              (hipStream_t)stream);
    if (rc) fail("swin_mlp_split (+ patch_unembed) launch rejected the shape (split storage, E = 192, HD = 768, NO = 160, HW % 128 == 0 only)");
    return rc;
}
int rs_op_softmax_rows(const float* s, void* out, long long nrows, int ncols, int out_prec, void* stream) {
    return rs_softmax_rows_launch(s, out, out_prec, nrows, ncols, ncols, ncols, (hipStream_t)stream);
}
int rs_op_vq(const float* z, const float* codebook_dev, float* zq, int32_t* idx, long long N, int NE, int D, void* stream) {
    return rs_vq_launch(z, codebook_dev, zq, idx, N, NE, D, (hipStream_t)stream);
}
int rs_op_nchw_to_nhwc(const float* in, void* out, int B, int C, int HW, int out_prec, void* stream) {
    return rs_nchw_to_nhwc_launch(in, out, out_prec, B, C, HW, C, 0, 1.f, (hipStream_t)stream);
}
int rs_op_nhwc_to_nchw(const void* in, float* out, int B, int C, int HW, int in_prec, void* stream) {
    return rs_nhwc_to_nchw_launch(in, in_prec, out, B, C, HW, C, 0, (hipStream_t)stream);
}
int rs_op_convert(const void* src, int src_prec, void* dst, int dst_prec, int C, long long npix, void* stream) {
    return rs_convert_launch(src, src_prec, dst, dst_prec, C, npix, (hipStream_t)stream);
}

}  // extern "C"
