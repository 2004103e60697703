/*
 * bndm_hip.h -- C ABI of libbndm_hip.so: the MI355X (gfx950) implementation of the bndm
 * sampling hot path (tiled blue-noise generator -> IADB/DDIM loop -> UNet2DModel forward).
 *
 * The reference (xchhuang/bndm) has no native boundary: its plug-in surface is a set of Python
 * call signatures.  Each entry point below names the reference call it stands in for
 * (file:line under /root/reference).  The Python layer in bndm_amd/ keeps those signatures and
 * forwards here through ctypes with tensor.data_ptr() and the current HIP stream.
 *
 * Conventions
 *   - plain pointers and sizes only; all device pointers are caller-owned (PyTorch allocations),
 *     the library owns only what it allocated behind an opaque handle;
 *   - every function returns 0 on success, a positive hipError_t or a negative BNDM_E_* code
 *     otherwise; bndm_last_error() gives a thread-local message; no C++ exception crosses the ABI;
 *   - all launches are asynchronous on the given stream (a hipStream_t passed as void*), no
 *     hidden synchronisation; one handle is used by one stream at a time.
 */
#ifndef BNDM_HIP_H
#define BNDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNDM_ABI_VERSION 1

#define BNDM_E_ARG      (-1)  /* invalid argument / unsupported shape (NotImplementedError upstream) */
#define BNDM_E_STATE    (-2)  /* handle not finalised / parameter missing                            */
#define BNDM_E_NOMEM    (-3)
#define BNDM_E_NODEVICE (-4)  /* no gfx950 device visible                                            */

int         bndm_abi_version(void);
const char *bndm_last_error(void);
/* number of visible HIP devices, name of device 0 copied to buf (may be NULL) */
int         bndm_device_info(char *buf, size_t buflen);

/* ------------------------------------------------------------------------------------------
 * Blue-noise generator: get_noise_v2 (bluenoise/get_noise_recent.py:23-196), branches
 *   32 px :77-99, 64 px :103-121, 128 px :126-164 + noise_padding :7-19,
 *   'gaussian' 128-px test re-layout :50-56.
 * Replaces torch.matmul(cov_mat_L, noise) + view/permute/cat/clone/lerp (K1-K4 in SURVEY 2.1).
 * ------------------------------------------------------------------------------------------ */
#define BNDM_Z_COLUMNS  0  /* z is [F, C, 64, 64]: 64 px (F=B), 32/128 px non-inplace draws (F=B / 4B) */
#define BNDM_Z_IMAGE32  1  /* z is x [B, C, 32, 32], replicated 2x2 periodically (:78-79)              */
#define BNDM_Z_IMAGE128 2  /* z is x [B, C, 128, 128], 2x2 tiles gathered tile-major (:131-133)        */

#define BNDM_NOISE_BLEND    0  /* gaussianBN / gaussianRN: bn*(1-a_b) + wn*a_b (:91,:116,:160)  */
#define BNDM_NOISE_PURE_BN  1  /* GBN: noise = noise_bn (:93,:118,:162)                          */
#define BNDM_NOISE_SCRAMBLE 2  /* 'gaussian', 128 px, test: re-layout only, L unused (:50-56)    */

/* bytes of scratch bndm_bluenoise needs for this shape (0 for BNDM_NOISE_SCRAMBLE) */
size_t bndm_bluenoise_workspace_bytes(int b_count, int C, int res);

/*
 * L        [4096,4096] f32 row-major (np.load(...)['x'], iadb_bn.py:83-86).
 * l_dense  0: caller guarantees L is exactly zero above the diagonal (only j<=i is read);
 *          1: full dense product, bit-for-bit the reference's semantics for any L.
 * z        white source, layout per z_layout; alpha [B_global] f32 or NULL.
 * noise / noise_bn / noise_wn : outputs [b_count, C, res, res] f32 (any may be NULL) for samples
 *          b_begin .. b_begin+b_count-1 of a global batch of B_global (128 px mixes tiles across
 *          the batch, SURVEY 8e caveat 1, so sharded ranks pass the global z and their range).
 */
int bndm_bluenoise(const float *L, int l_dense, const float *z, int z_layout, const float *alpha,
                   float *noise, float *noise_bn, float *noise_wn,
                   int B_global, int b_begin, int b_count, int C, int res, int mode,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Sampler steps (K7, K13, K14)
 * ------------------------------------------------------------------------------------------ */
/* x += da*d[:, :C] (+ dg*d[:, C:2C] when Cout == 2C).  utils.py:216-226, iadb_bn.py:323-344,
 * latent_iadb_bn_diffusers.py:107-117.  x [B,C,HW] f32, d [B,Cout,HW] f32. */
int bndm_iadb_step(float *x, const float *d, float da, float dg, int B, int C, int Cout, int HW,
                   void *stream);

/* Training-time noise injection, the arithmetic after get_noise_v2(..., 'train', inplace=False)
 * (iadb_bn.py:881): forward blend x_alpha = alpha*x0 + (1-alpha)*x1 (iadb_bn.py:915; x1 = data, x0 = noise;
 * IADBScheduler.add_noise, latent_iadb_bn_diffusers.py:128-133,610) and the regression targets
 * tar1 = x1 - x0, tar2 = alpha_prev*(noise_bn - noise_wn), tar = tar1 + tar2 (iadb_bn.py:946-956,
 * latent_iadb_bn_diffusers.py:617-620).  All tensors [B, per_sample] f32 on the device, alpha / alpha_prev [B];
 * noise_bn == noise_wn == NULL selects the 'gaussian' / 'GBN' target (tar = tar1).  Any output may be NULL. */
int bndm_iadb_train_targets(const float *x0, const float *x1, const float *noise_bn, const float *noise_wn,
                            const float *alpha, const float *alpha_prev, float *x_alpha, float *tar1,
                            float *tar2, float *tar, int B, size_t per_sample, void *stream);

/* DDIMScheduler.step(...).prev_sample (ddim_diffusers.py:680): eps-prediction, eta 0,
 * x0 = clamp((x - sqrt(1-a_t) eps)/sqrt(a_t), +-clip); x' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps.
 * clip <= 0 disables clipping.  In place on x. */
int bndm_ddim_step(float *x, const float *eps, float sqrt_at, float sqrt_1m_at, float sqrt_ap,
                   float sqrt_1m_ap, float clip, size_t n, void *stream);

/* clamp((x+1)/2,0,1)*255 -> u8, NCHW -> NHWC.  rounding 0: truncate (iadb_bn.py:815-816),
 * 1: round-half-even (ddim_diffusers.py:687-688, latent_iadb_bn_diffusers.py:539-540). */
int bndm_export_u8(const float *x, uint8_t *out, int B, int C, int HW, int rounding, void *stream);

/* ------------------------------------------------------------------------------------------
 * UNet2DModel engine: diffusers.UNet2DModel as constructed at iadb_bn.py:205-282,
 * utils.py:7-84, ddim_diffusers.py:377-453, latent_iadb_bn_diffusers.py:337-372 and called at
 * iadb_bn.py:319 (model(x, alpha, return_dict=False)[0]) / ddim_diffusers.py:679 (.sample).
 * ------------------------------------------------------------------------------------------ */
typedef struct bndm_unet bndm_unet;

#define BNDM_MAX_LEVELS 8
#define BNDM_DTYPE_F16  0
#define BNDM_DTYPE_BF16 1
#define BNDM_DTYPE_F32  2  /* fp32-compute verification mode: fp32 NCHW, plain FMA kernels, max_batch <= 8.
                            * The reference's own arithmetic (iadb_bn.py:304-344 samples in fp32, no autocast). */

typedef struct bndm_unet_config {
    int in_channels;
    int out_channels;
    int resolution;                              /* H == W of `sample`, power of two          */
    int num_levels;                              /* len(block_out_channels)                   */
    int block_out_channels[BNDM_MAX_LEVELS];
    int down_attn[BNDM_MAX_LEVELS];              /* 1: AttnDownBlock2D at this index          */
    int up_attn[BNDM_MAX_LEVELS];                /* 1: AttnUpBlock2D at this index            */
    int layers_per_block;                        /* 2 in every reference config               */
    int dtype;                                   /* storage/MFMA input type of activations    */
    int max_batch;                               /* workspaces are sized for this batch       */
} bndm_unet_config;

int  bndm_unet_create(bndm_unet **out, const bndm_unet_config *cfg);
void bndm_unet_destroy(bndm_unet *h);

/* number of parameters / the i-th state-dict key (diffusers naming) and its element count */
int  bndm_unet_num_params(const bndm_unet *h);
int  bndm_unet_param_info(const bndm_unet *h, int index, char *name, size_t name_len, int64_t *numel);

/* model.load_state_dict(...) (iadb_bn.py:714): one tensor, host f32, PyTorch layout
 * (conv OIHW, linear [out,in]); converted/re-laid out on upload. */
int  bndm_unet_load_param(bndm_unet *h, const char *name, const float *host_data, int64_t numel);
/* all parameters present -> pack derived tables; must precede forward */
int  bndm_unet_finalize(bndm_unet *h);

/* The launch list of one forward, fixed at finalize (it depends on max_batch: tile variants are chosen for the
 * handle's batch size).  kernel: family + tile variant ("conv_t32<TH=16>", "conv_igemm", "gn_small", ...); label: the
 * layer it computes (diffusers module path).  Lets a test assert which kernels a configuration runs; no reference
 * counterpart (the reference's launch list is whatever eager PyTorch dispatches at iadb_bn.py:319). */
int  bndm_unet_num_ops(const bndm_unet *h);
int  bndm_unet_op_info(const bndm_unet *h, int index, char *kernel, size_t kernel_len, char *label,
                       size_t label_len, double *flops_per_sample);

/* sample [B,Cin,H,W] f32 (device), timesteps [B] f32 (device), out [B,Cout,H,W] f32 (device) */
int  bndm_unet_forward(bndm_unet *h, const float *sample, const float *timesteps, float *out,
                       int B, void *stream);

/* Whole IADB loop (utils.py:196-226): for s = 0..nb_step-1 (t = nb_step-1-s):
 *   d = unet(x, t_in[s]);  x += da[s]*d[:, :C] + dg[s]*d[:, C:]      (dg ignored when Cout == C)
 * t_in/da/dg are host arrays of nb_step floats (precomputed schedule tables, K6).
 * extra_in: optional [B, Cin-C, H, W] conditioning concatenated to x each step
 * (sample_iadb_conditional, iadb_bn.py:406), else NULL.
 * snapshots: optional device buffer [n_snap, B, C, H, W]; x is copied there after every step s
 * with snap_mask[s] != 0 (x_all, utils.py:229-235). */
int  bndm_unet_sample_iadb(bndm_unet *h, float *x, const float *extra_in, int B, int C, int nb_step,
                           const float *t_in, const float *da, const float *dg,
                           const uint8_t *snap_mask, float *snapshots, void *stream);

/* Whole DDIM loop (ddim_diffusers.py:674-681); coef is host [nb_step][5] =
 * {t, sqrt_at, sqrt_1m_at, sqrt_ap, sqrt_1m_ap}. */
int  bndm_unet_sample_ddim(bndm_unet *h, float *x, int B, int nb_step, const float *coef, float clip,
                           void *stream);

/* per-forward kernel time of the last bndm_unet_forward/sample call is not tracked here;
 * use bndm_unet_profile to time the named stage classes with HIP events on `stream`. */
typedef struct bndm_unet_profile_t {
    float ms_total;        /* one forward, events on the launch stream                     */
    float ms_conv;         /* implicit-GEMM conv kernels (dominant, MFMA-bound)            */
    int   conv_launches;
    double conv_flops;     /* algorithmic 2*MAC of those launches                          */
    int   launches;        /* all kernel launches of one forward                           */
    /* the single dominant kernel (conv_t32 with 256-pixel tiles): the roofline line of bench.py */
    float  ms_dom;         /* sum of its launch durations in one forward                   */
    int    dom_launches;
    double dom_flops;      /* algorithmic 2*MAC of those launches                          */
    double dom_bytes;      /* algorithmic HBM bytes: inputs + residual + output + weights  */
} bndm_unet_profile_t;
int  bndm_unet_profile(bndm_unet *h, const float *sample, const float *timesteps, float *out, int B,
                       int iters, bndm_unet_profile_t *prof, void *stream);

/* ------------------------------------------------------------------------------------------
 * AutoencoderKL decoder: vae.decode((x / 0.18215).half()).sample as called at
 * latent_iadb_bn_diffusers.py:185-191,531-533 (vae = AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-mse"),
 * :70-71).  The handle is the same opaque type as the UNet's: bndm_unet_num_params / param_info / load_param /
 * finalize / destroy apply; state-dict keys are diffusers' ("post_quant_conv.*", "decoder.*").
 * ------------------------------------------------------------------------------------------ */
typedef struct bndm_vae_config {
    int latent_channels;                         /* 4                                              */
    int out_channels;                            /* 3                                              */
    int latent_resolution;                       /* H == W of the latent: 16, 32 or 64             */
    int num_levels;                              /* len(block_out_channels)                        */
    int block_out_channels[BNDM_MAX_LEVELS];     /* encoder order, (128, 256, 512, 512)            */
    int layers_per_block;                        /* 2 (the decoder runs layers_per_block + 1)      */
    int dtype;                                   /* BNDM_DTYPE_F16 / BNDM_DTYPE_BF16               */
    int max_batch;                               /* <= 8                                           */
} bndm_vae_config;

int  bndm_vae_decoder_create(bndm_unet **out, const bndm_vae_config *cfg);
/* latents [B, latent_channels, r, r] f32 (device), ALREADY divided by the scaling factor;
 * out [B, out_channels, r << (num_levels-1), same] f32 (device) */
int  bndm_vae_decode(bndm_unet *h, const float *latents, float *out, int B, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BNDM_HIP_H */
