/* resshift_hip.h — C ABI of the MI355X-native ResShift sampling engine (libresshift_hip.so).
 *
 * The reference (zsyOAOA/ResShift) is pure Python/PyTorch and has no FFI of its own; its only
 * extension point is the YAML `target:` string resolved by utils/util_common.py:19-29.  The entry
 * points below are therefore exactly what a ctypes binding of the reference's hot path needs:
 * each one replaces the body of one reference call (file:line cited per function), takes plain
 * pointers and sizes, and never sees a torch type.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (rs_last_error() has the text);
 *   - "dev" pointers are device pointers valid on the current HIP device (e.g. torch
 *     Tensor.data_ptr()); the engine never frees caller memory;
 *   - user-facing image / latent tensors are NCHW fp32 like the reference's; the NHWC fp16/fp32
 *     working layout is internal;
 *   - work is enqueued on the caller's hipStream_t (pass torch's current stream); one engine per
 *     device, not thread-safe;
 *   - precision: RS_PREC_F16 = fp16 storage / fp32 accumulate MFMA, RS_PREC_F32 = fp32 storage /
 *     exact fp32 MFMA, RS_PREC_SPLIT = (hi, lo) fp16 pair storage (x = hi + lo * 2^-11, 4 bytes per
 *     element) / three fp16 MFMAs per product, fp32 accumulate: fp32-class results at 1/3 of the
 *     fp16 matrix rate (the precision that keeps the VQ codes of ldm/modules/vqvae/quantize.py:276-285).
 */
#ifndef RESSHIFT_HIP_H
#define RESSHIFT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_PREC_F16 0
#define RS_PREC_F32 1
#define RS_PREC_SPLIT 2
#define RS_MAX_LEVELS 8
#define RS_MAX_STEPS 64

typedef struct rs_engine rs_engine;

/* models/unet.py:632-657 (UNetModelSwin.__init__ arguments that shape the network) */
typedef struct rs_unet_config {
    int image_size, in_channels, model_channels, out_channels;
    int n_levels;
    int channel_mult[RS_MAX_LEVELS];
    int num_res_blocks[RS_MAX_LEVELS];
    int n_attn_res;
    int attention_resolutions[RS_MAX_LEVELS];
    int swin_depth, swin_embed_dim, window_size, num_heads;
    float mlp_ratio;
    int cond_lq, cond_mask, lq_size;
} rs_unet_config;

/* ldm/models/autoencoder.py:13-26 + ldm/modules/diffusionmodules/model.py:452-456,550-554 (ddconfig) */
typedef struct rs_ae_config {
    int ch, n_levels;
    int ch_mult[RS_MAX_LEVELS];
    int num_res_blocks[RS_MAX_LEVELS];   /* per level (an int in the YAML is broadcast, model.py:464-468) */
    int in_channels, out_ch, z_channels, embed_dim, n_embed, resolution;
    int n_attn_res;
    int attn_resolutions[RS_MAX_LEVELS];
} rs_ae_config;

typedef struct rs_config {
    rs_unet_config unet;
    rs_ae_config ae;
    int has_unet;     /* build the UNetModelSwin graph   */
    int has_ae;       /* build the VQModelTorch graph    */
    int enable_f16;   /* pack fp16 weights  */
    int enable_f32;   /* pack fp32 weights (exact mode) */
    int enable_split; /* pack (hi, lo) fp16 pair weights (RS_PREC_SPLIT) */
} rs_config;

/* Arguments of one full sampling call: gaussian_diffusion.py:367-472 (p_sample_loop) */
typedef struct rs_sample_args {
    const float* y;        /* dev, [B,3,h,w] LR image in [-1,1]                     */
    const float* mask;     /* dev, [B,1,h,w] or NULL (inpainting, sampler.py:140)    */
    const float* noise;    /* dev, [steps+1][B,Cz,hz,wz]: prior noise then one per step in loop order (t=T-1..0) */
    float* out;            /* dev, [B,3,h*sf,w*sf] decoded image (not clamped)       */
    float* z_out;          /* dev, optional [B,Cz,hz,wz] final latent before VQ      */
    int32_t* idx_out;      /* dev, optional [B*hz*wz] VQ code indices                */
    int B, h, w, sf, steps;
    /* per-step scalars, index = timestep t (float64 numpy -> float, gaussian_diffusion.py:143-161,602) */
    float inv_std[RS_MAX_STEPS];   /* 1/sqrt(eta_t*kappa^2+1)              (_scale_input)       */
    float coef1[RS_MAX_STEPS];     /* eta_{t-1}/eta_t                      (posterior_mean_coef1) */
    float coef2[RS_MAX_STEPS];     /* alpha_t/eta_t                        (posterior_mean_coef2) */
    float sigma[RS_MAX_STEPS];     /* exp(0.5*posterior_log_variance_clipped[t])              */
    int tmap[RS_MAX_STEPS];        /* respace.py:61-70 timestep_map                           */
    float prior_scale;             /* kappa*sqrt_eta_{T-1}                 (prior_sample)       */
    float scale_factor;            /* diffusion.params.scale_factor                           */
    int prec_encode, prec_decode;
    int prec_unet[RS_MAX_STEPS];   /* precision of the UNet call at timestep t */
    void* stream;                  /* hipStream_t */
} rs_sample_args;

/* ---- lifetime ------------------------------------------------------------------------------ */
rs_engine* rs_create(const rs_config* cfg);
void rs_destroy(rs_engine* e);
const char* rs_last_error(void);

/* ---- weights: replaces utils/util_net.py:86-98 (reload_model) + sampler.py:108-112 ----------- */
/* hand over one state_dict entry (fp32 host memory, reference key name, reference shape) */
int rs_load_tensor(rs_engine* e, const char* state_dict_key, const float* host, const int64_t* shape, int ndim);
/* size of the packed device blob for this config; layout is a pure function of the config */
size_t rs_weight_bytes(rs_engine* e);
/* caller-owned device storage for the blob (so the caller can RCCL-broadcast it as one message) */
int rs_bind_weight_blob(rs_engine* e, void* dev, size_t bytes);
/* pack everything loaded so far into the bound blob (GEMM-ready [Cout][kh][kw][Cin] fp16/fp32,
 * expanded relative-position bias tables, codebook, ...); only the broadcasting rank needs to call it */
int rs_pack_weights(rs_engine* e);
/* broadcast the bound blob from rank `root` over the HOST's RCCL communicator (`rccl_comm`: an ncclComm_t; one ncclBroadcast of
 * rs_weight_bytes() bytes, in place, on `stream`): what replaces the reference's per-rank checkpoint load (sampler.py:66-77, utils/util_net.py
 * reload_model) for a host without torch.distributed.  RCCL is dlopen'ed at call time - no link dependency.  Every rank calls rs_weights_ready()
 * afterwards. */
int rs_bcast_weights(rs_engine* e, void* rccl_comm, int root, void* stream);
/* tell the engine the blob content is valid (after pack or after a broadcast) */
int rs_weights_ready(rs_engine* e);

/* ---- network calls ------------------------------------------------------------------------- */
/* models/unet.py:865-895 UNetModelSwin.forward(x, timesteps, lq, mask).  x [B,Cz,H,W], lq [B,3,Hl,Wl],
 * mask [B,1,Hl,Wl] or NULL, out [B,out_channels,H,W]; t_host = B timestep values on the host. */
int rs_unet_forward(rs_engine* e, const float* x, const int* t_host, const float* lq, const float* mask, float* out,
                    int B, int H, int W, int Hl, int Wl, int prec, void* stream);
/* ldm/models/autoencoder.py:28-31 VQModelTorch.encode: img [B,3,H,W] -> z [B,embed_dim,H/f,W/f] */
int rs_vq_encode(rs_engine* e, const float* img, float* z, int B, int H, int W, int prec, void* stream);
/* ldm/models/autoencoder.py:33-40 VQModelTorch.decode: z [B,embed_dim,h,w] -> img [B,out_ch,h*f,w*f] */
int rs_vq_decode(rs_engine* e, const float* z, float* img, int32_t* idx_out, int B, int h, int w, int force_not_quantize,
                 int prec, void* stream);
/* F.interpolate(y, scale_factor=sf, mode='bicubic') — gaussian_diffusion.py:503-504; NCHW in / NCHW out */
int rs_bicubic(rs_engine* e, const float* y, float* out, int B, int C, int H, int W, int sf, void* stream);
/* the whole loop: encode_first_stage -> prior_sample -> steps x (UNet + posterior update) -> decode */
int rs_sample(rs_engine* e, const rs_sample_args* a);
/* fp32 y = a*x + b*z + c*n on device (posterior mean / prior sample for the step-wise API) */
int rs_axpbypcz(const float* x, const float* z, const float* n, float* y, float a, float b, float c, long long count, void* stream);

/* overlap-average tiling of large images (utils/util_image.py:889-979 ImageSpliterTh.update / .gather): NCHW fp32,
 * acc[b,c,h0:h0+th,w0:w0+tw] += tile, count[h0:h0+th,w0:w0+tw] += 1; finalize divides acc by count in place */
int rs_tile_accumulate(float* acc, float* count, const float* tile, int B, int C, int H, int W, int h0, int w0, int th, int tw,
                       void* stream);
int rs_tile_finalize(float* acc, const float* count, int B, int C, int H, int W, void* stream);
/* fp32 planes [planes][H][W] -> [planes][Ho][Wo]: out[i][j] = scale * in[refl(h0 + i)][refl(w0 + j)], refl(i) = i < n ? i : 2(n-1) - i.
 * The host mirror's data movement on the device: reflect padding of the LQ batch (sampler.py:130-138), the tile crop of the
 * tiled path (utils/util_image.py:946-952; the window must then lie inside the plane) and the latent scaling of
 * encode_first_stage (models/gaussian_diffusion.py:514).  -2 when the window overhangs the plane by a full plane size or more. */
int rs_window_copy(const float* in, float* out, long long planes, int H, int W, int h0, int w0, int Ho, int Wo, float scale, void* stream);

/* uint8 pre / post processing on the device.
 * rs_u8_to_input:  interleaved uint8 [B,H,W,C] -> planar fp32 [B,C,H,W] in [-1,1]  ((v/255 - 0.5)/0.5; replaces
 *                  datapipe/datasets.py:59-63 ToTensor + Normalize on the host)
 * rs_output_to_u8: planar fp32 [-1,1] -> interleaved uint8: x*0.5+0.5, optional inpainting blend with the LQ input
 *                  sr*m + lq*(1-m) (sampler.py:218-222; lq and mask both null or both set, mask is [B,1,H,W] in [-1,1]),
 *                  clamp, *255, round-half-even, RGB->BGR when `bgr` (utils/util_image.py:245-269 tensor2img) */
int rs_u8_to_input(const void* src_u8_nhwc, float* dst_f32_nchw, int B, int H, int W, int C, void* stream);
int rs_output_to_u8(const float* sr_f32_nchw, const float* lq_f32_nchw, const float* mask_f32_n1hw, void* dst_u8_nhwc, int B, int H,
                    int W, int C, int bgr, void* stream);

/* ---- introspection --------------------------------------------------------------------------- */
/* bytes of scratch arena currently allocated; number of kernel launches issued by the last call */
size_t rs_arena_bytes(rs_engine* e);
long long rs_last_launch_count(rs_engine* e);
/* profiling of the MFMA implicit-GEMM kernel family: when enabled every igemm launch of the next call is
 * bracketed by hipEvents on the launch stream.  rs_profile_get fills out[9] = {fp16-input igemm FLOPs,
 * fp32-input igemm FLOPs, summed igemm kernel milliseconds, igemm launch count, algorithmic HBM bytes (every
 * operand and result counted once), split-input igemm FLOPs, GroupNorm kernel milliseconds, GroupNorm count, GroupNorm
 * algorithmic bytes (input read once + output written once)} of the last call (counts are always maintained; the time is 0 unless
 * profiling was on).  The cost of an empty event pair, measured on the same stream, is subtracted from every
 * bracket so that the sum is the kernels' own duration. */
int rs_profile_enable(rs_engine* e, int on);
int rs_profile_get(rs_engine* e, double* out9);
/* the same per kernel family of the MFMA path, in this order: halo conv fp16 (igemm4), halo conv split storage, implicit GEMM fp16
 * (igemm2 / igemm3 / igemm), implicit GEMM split storage, implicit GEMM fp32 (exact), fused qkv + window attention + projection,
 * fused Swin MLP, and the split-storage variants of those two.  out[3 f + 0] = algorithmic FLOPs, [3 f + 1] = kernel milliseconds,
 * [3 f + 2] = launches; returns the number of families (9) or -1 when cap < 27. */
int rs_profile_families(rs_engine* e, double* out, int cap);
/* debug trace (tests only): when enabled, the next network call records named intermediate activations
 * (scratch is not recycled while enabled); fetch converts entry i to NCHW fp32 into caller memory. */
/* text table of the last profiled call: per (part, kernel family, M, N, K) launch shape of the MFMA family - launches, summed kernel ms
 * (hipEvents on the launch stream), algorithmic flops - and the wall ms of the encoder / UNet / decoder parts (measurement, d of SURVEY 8:
 * where a pass's time goes per reference module - ldm/modules/diffusionmodules/model.py Encoder / Decoder, models/unet.py UNetModelSwin).
 * Returns the bytes needed incl. the terminating 0; copies at most cap. */
int rs_profile_shapes(rs_engine* e, char* buf, int cap);
int rs_debug_enable(rs_engine* e, int on);
int rs_debug_count(rs_engine* e);
int rs_debug_info(rs_engine* e, int i, char* name, int name_cap, int* dims_bchw);
int rs_debug_fetch(rs_engine* e, int i, float* out_nchw_dev, void* stream);

/* ---- op-level entry points (used by tests/ to check each kernel against torch on its own) ---- */
/* NHWC conv / linear through the MFMA implicit GEMM or the direct kernels (auto-selected).
 * x0 [B,Hs,Ws,C0] (+ optional x1 [B,Hs,Ws,C1] concatenated on C), w_ref in the reference layout
 * [Cout][C0+C1][KH][KW] fp32 on the HOST (packed internally), bias fp32 host or NULL, res/out NHWC. */
int rs_op_conv2d(const void* x0, const void* x1, const float* w_ref_host, const float* bias_host, const void* res, void* y,
                 int B, int Hs, int Ws, int C0, int C1, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho,
                 int Wo, int up, int act, int in_prec, int out_prec, int force_direct, void* stream);
/* timing probe for one implicit-GEMM conv shape: device-resident NHWC input, weights already packed [Cout][KH*KW*Cin]
 * in the input precision; runs `reps` launches between two hipEvents and returns the average ms per launch. */
int rs_op_conv2d_bench(const void* x0, const void* w_packed_dev, const float* bias_dev, const void* res, void* y, int B, int Hs,
                       int Ws, int Cin, int Cout, int KH, int KW, int stride, int pad, int Ho, int Wo, int up, int act, int in_prec,
                       int out_prec, int reps, float* ms_out, void* stream);
/* halo-tile 3x3 conv with the GroupNorm affine + activation of its input fused in (igemm4.hip): y = conv3x3(act_in(x * coef[b][0][c]
 * + coef[b][1][c])) (+res); x / res / y NHWC device tensors in `prec` storage (RS_PREC_F16 or RS_PREC_SPLIT), coef_dev [B][2][Cin]
 * fp32 device (null: plain conv), act_in 0 / 2 (none / SiLU), weights [Cout][Cin][3][3] fp32 host.  `ystats_dev` (may be null):
 * [B][H*W / rs_op_conv3x3_halo_stats_px(...)][Cout][2] fp32 device, receives the per-(image, 256-pixel slab, channel) sum / sum of squares of the stored
 * output (the GroupNorm statistics the kernel - or, for its split-K launches on the 16x16 / 8x8 planes, the reduce kernel - leaves
 * for the consuming GroupNorm).  Returns an error when the shape is not eligible for that kernel. */
int rs_op_conv3x3_halo(const void* x, const float* coef_dev, int act_in, const float* w_ref_host, const float* bias_host, const void* res,
                       void* y, int B, int H, int W, int Cin, int Cout, int prec, float* ystats_dev, void* stream);
/* the same layer on the Winograd F(2x2,3x3) kernel (wino.hip; RS_PREC_SPLIT tensors only): U = G g G^T packed on the host, one checked launch;
 * reps > 0: `reps` more launches between two hipEvents, *ms_out = average ms per launch.  ystats_dev (may be null): [B][H*W / 128][Cout][2]
 * (one slab per 8 x 16 pixel tile).  Replaces the reference's nn.Conv2d(3x3, padding 1) behind GroupNorm + SiLU (models/unet.py:128-147,186-206; ldm/modules/diffusionmodules/
 * model.py:100-149).  Returns an error when the shape is not eligible (Cin % 32, Cout % 64, planes that tile by 8 x 16, Cin <= 640). */
int rs_op_conv3x3_wino(const void* x, const float* coef_dev, int act_in, const float* w_ref_host, const float* bias_host, const void* res,
                       void* y, int B, int H, int W, int Cin, int Cout, float* ystats_dev, int reps, float* ms_out, void* stream);
/* pixels per statistics slab rs_op_conv3x3_halo uses for this shape (0: shape not eligible / no statistics): ystats_dev is
 * [B][H*W / slab][Cout][2] (256- or 128-pixel tiles of the kernel variant, or the reduce kernel's slabs for its split-K launches) */
int rs_op_conv3x3_halo_stats_px(int B, int H, int W, int Cin, int Cout, int prec);
/* batched NT GEMM: y[z][m][n] = scale * sum_k a[z][m][k] * b[z][n][k]  (+bias[n]) */
int rs_op_gemm_nt(const void* a, const void* b, const float* bias_dev, void* y, int nz, int M, int N, int K, float scale,
                  int in_prec, int out_prec, void* stream);
int rs_op_groupnorm(const void* x, void* y, const float* gamma_host, const float* beta_host, const float* film_dev, int B, int HW,
                    int C, int groups, float eps, int act, int prec, void* stream);
int rs_op_window_attention(const void* qkv, void* out, const float* bias_table_host /*[225][heads]*/, int B, int H, int W,
                           int heads, int shift, int prec, void* stream);
/* fused qkv projection + window attention (fp16, 6 heads of 32): x [B,H,W,192] normalised tokens, wqkv [576][192] fp16 device,
 * bqkv [576] fp32 device, table as in rs_op_window_attention (host); out [B,H,W,192].  With wproj_dev ([192][192] fp16) and
 * bproj_dev the output projection is fused as well and `res` (fp16 [B,H,W,192], may be null) is added: out = res + proj(attn) */
int rs_op_window_attention_qkv(const void* x, const void* wqkv_dev, const float* bqkv_dev, const void* wproj_dev, const float* bproj_dev,
                               const void* res, void* out, const float* table_host, int B, int H, int W, int heads, int shift, void* stream);
/* the same on split storage (RS_PREC_SPLIT): x / res / out are (hi, lo) pair tensors, wqkv_dev [576][192 hi | 192 lo] and wproj_dev
 * [192][192 hi | 192 lo] fp16 on the device; `xcoef_dev` (may be null) is a GroupNorm affine [B][2][192] applied to x on the fly
 * (models/swin_transformer.py:85-145,238-277 in one launch, win_attn_split.hip) */
int rs_op_window_attention_qkv_split(const void* x, const void* wqkv_dev, const float* bqkv_dev, const void* wproj_dev, const float* bproj_dev,
                                     const void* res, void* out, const float* table_host, const float* xcoef_dev, int B, int H, int W, int heads,
                                     int shift, void* stream);
/* streaming attention of the autoencoder's AttnBlock (ldm/modules/diffusionmodules/model.py:179-203), fp16 device tensors: q, k
 * [nz][T][C], vt [nz][C][T] (v transposed, without its bias), bv_dev [C] fp32 (v bias, may be null), o [nz][T][C] =
 * softmax(q k^T / sqrt(C)) v + bv; C in {128, 256, 512}, T a multiple of 128 (ae_attn.hip) */
int rs_op_ae_flash_attention(const void* q, const void* k, const void* vt, const float* bv_dev, void* o, int nz, int T, int C, void* stream);
/* the same on split storage (RS_PREC_SPLIT tensors of (hi, lo) fp16 pairs: q, k, o records [C hi | C lo] per token, vt rows [T hi | T lo] per
 * channel); C = 512, T a multiple of 64 (ae_attn_split.hip; the reference's memory-efficient AttnBlock is model.py:205-268) */
int rs_op_ae_flash_attention_split(const void* q, const void* k, const void* vt, const float* bv_dev, void* o, int nz, int T, int C, void* stream);
/* fused Swin MLP, fp16 device tensors: y[M][E] = res + fc2(GELU(fc1(x))) with fc1 weights [HD][E], fc2 weights [E][HD]
 * (row-major fp16, device), fp32 biases; res may be null (models/swin_transformer.py:17-33,279) */
int rs_op_swin_mlp(const void* x, const void* w1_dev, const float* b1_dev, const void* w2_dev, const float* b2_dev, const void* res, void* y,
                   int M, int E, int HD, void* stream);
/* the same on split storage (RS_PREC_SPLIT tensors of (hi, lo) fp16 pairs): weights packed [rows][K hi | K lo] fp16 on the device */
int rs_op_swin_mlp_split(const void* x, const void* w1_dev, const float* b1_dev, const void* w2_dev, const float* b2_dev, const void* res, void* y,
                         int M, int E, int HD, void* stream);
/* the last Swin block of a BasicLayer + its patch_unembed (models/swin_transformer.py:279,515,521-528) in one launch, split storage:
 * y[M][NO] = Wu (x + fc2(GELU(fc1(x a + d))) + b2) + bu with the GroupNorm affine (a, d) = xcoef_dev [M / HW][2][E]; w2cat_dev = the product
 * matrix [NO][HD + E] = [Wu W2 | Wu] packed [rows][K hi | K lo], bcat_dev = Wu b2 + bu; E = 192, HD = 768, NO = 160, HW % 128 == 0 */
int rs_op_swin_mlp_split_unembed(const void* x, const float* xcoef_dev, const void* w1_dev, const float* b1_dev, const void* w2cat_dev, const float* bcat_dev,
                                 void* y, int M, int HW, int E, int HD, int NO, void* stream);
int rs_op_softmax_rows(const float* s, void* out, long long nrows, int ncols, int out_prec, void* stream);
int rs_op_vq(const float* z, const float* codebook_dev, float* zq, int32_t* idx, long long N, int NE, int D, void* stream);
int rs_op_nchw_to_nhwc(const float* in, void* out, int B, int C, int HW, int out_prec, void* stream);
int rs_op_nhwc_to_nchw(const void* in, float* out, int B, int C, int HW, int in_prec, void* stream);
/* storage conversion of an NHWC tensor [npix][C] between any two of RS_PREC_F16 / RS_PREC_F32 / RS_PREC_SPLIT (tests feed and
 * read split-storage tensors through it) */
int rs_op_convert(const void* src, int src_prec, void* dst, int dst_prec, int C, long long npix, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RESSHIFT_HIP_H */
