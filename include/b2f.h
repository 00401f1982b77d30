/*
 * b2f.h -- C ABI of libb2f.so: the MI355X-native (gfx950) drop-in for the
 * back2future `init()` -> `computeFlow(im1, im2, im3)` hot path.
 *
 * The reference has no C ABI for this path: its native boundary is the Torch7
 * Lua-C extension protocol (luaL_Reg tables on the tensor metatable,
 * /root/reference/extras/stnbhwd/BilinearSamplerBHWD.cu:423-435, init.cu:11-18)
 * and, above it, the Lua function pair exported by back2future.lua:45,97-130.
 * Each entry point below cites the reference interface it replaces.  The
 * LuaJIT-ffi / ctypes bindings that call this header are shown in
 * INTEGRATION.md.
 *
 * Conventions: every call returns 0 on success, non-zero on error with a
 * thread-local message readable through b2f_last_error() (the Lua shim turns
 * it into error(msg), mirroring THError at BilinearSamplerBHWD.cu:151-156).
 * Plain pointers and sizes only; the caller owns every buffer it passes in;
 * the library owns device memory inside the opaque context.  A context is
 * bound to one GPU; several GPUs of a node are driven either by one process per GPU
 * (one context per rank, the host broadcasts the flat weight buffer over RCCL, see
 * b2f_weights_device) or by one b2f_multi inside one process (b2f_init_multi below).
 * "host" pointers are CPU memory, "dev" pointers are HIP device memory on the
 * context's GPU.  `stream` is a hipStream_t passed as void* (NULL = the
 * context's own stream).
 */
#ifndef B2F_H
#define B2F_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define B2F_API __attribute__((visibility("default")))
#else
#define B2F_API
#endif

typedef struct b2f_ctx b2f_ctx;

/* Error text of the last failing call on this thread ("" if none). */
B2F_API const char *b2f_last_error(void);

/* Library/ABI version (major*1000 + minor). */
B2F_API int b2f_version(void);

/* ---- life cycle: replaces back2future.init(opt) (back2future.lua:97-129) ----
 * name_or_path:
 *   "Ours-Hard" | "Ours-Soft-ft-KITTI" | "Ours-Soft-ft-Sintel"
 *        -> models/RoamingImages_H.t7 | _H_KITTI_S.t7 | _H_Sintel_S.t7 relative to
 *           the current directory, exactly as back2future.lua:100-110 (error if absent)
 *   a path ending in ".t7"    -> Torch7 serialized nn.gModule / DataParallelTable
 *   a path ending in ".b2fw"  -> flat fp32 blob in this repo's canonical order
 *   "random:hard[:seed[:gain]]" | "random:soft[:seed[:gain]]"
 *        -> the pwc.lua architecture with deterministic random weights
 *           (nn.SpatialConvolution:reset() distribution), for synthetic benchmarks.
 * device: HIP device ordinal.                                                    */
B2F_API int b2f_init(const char *name_or_path, int device, b2f_ctx **out);
/* Same with the graph shape of createModelMulti(opt) (models/pwc.lua:88-121) given explicitly, for models other
 * than the shipped ones: graph_opts = "win=5,levels=4" (createModelMulti(nil), pwc.lua:88), or any subset of
 * win (pwc_ws), levels, skip (pwc_skip, >= 0), two_frame, sum_cvs (pwc_sum_cvs), residual, occ_input,
 * rescale_flow, flownet_factor, siamese (pwc_siamese); NULL / "" = the shipped graph (opts.lua:83-98) -- or, for a ".t7" file, whatever
 * shape the file holds: win is read from its nn.CostVolMulti nodes (CostVolMulti.lua:26-37), levels and skip from the
 * convUnits / decoders in its node list (pwc.lua:136,237), the other options stay at their defaults; with graph_opts
 * given the file must be that graph.  frames = 3 is fixed; a pwc_siamese = 0 model (no convUnits: decoders of every
 * level have the same shapes) is not read from .t7, only from flat weights.  Weights: "random:hard|soft[:seed[:gain]]", a ".t7" file or a .b2fw blob in the canonical order of that graph
 * (feature units l = 2..levels -- from l = 1 with skip = 0, none with siamese = 0 --, then l = levels..skip+1 {occ, flow,
 * [past-flow] decoder}).  Non-shipped shapes run
 * on a generic, untuned executor (every Lua node its own kernels); H and W of b2f_forward must then be multiples
 * of 2^(levels-1), computeFlow keeps the reference's /64 rounding.                                            */
B2F_API int b2f_init_ex(const char *name_or_path, int device, const char *graph_opts, b2f_ctx **out);
B2F_API void b2f_destroy(b2f_ctx *ctx);

/* levels (7), cost-volume window (9), past_flow (0 Hard / 1 Soft), number of
 * tensors in the model:forward output table ((levels - skip) x 4 | 5: 20 / 25 for the shipped
 * models, pwc.lua:459-489), #params. */
B2F_API int b2f_info(const b2f_ctx *ctx, int *levels, int *win, int *past_flow, int *n_outputs,
             long long *n_params);

/* ---- weights (the .t7 payload of back2future.lua:113) ----
 * Canonical flat order: feature units l=2..7 {conv1.w,b,conv2.w,b};
 * then l=7..3 {occ decoder, flow decoder, [past-flow decoder]} x 6 x {w,b};
 * every w is Co x Ci x 3 x 3 as in nn.SpatialConvolution.                        */
B2F_API long long b2f_param_count(int past_flow);
/* Host-only, no GPU needed: fills out[n] with the deterministic random init. */
B2F_API int b2f_random_weights(unsigned long long seed, int past_flow, float gain, float *out,
                       long long n);
B2F_API int b2f_set_weights(b2f_ctx *ctx, const float *host_flat, long long n);
B2F_API int b2f_get_weights(b2f_ctx *ctx, float *host_flat, long long n);
/* Device address of the flat canonical buffer, so that the host can broadcast it in
 * place with RCCL (torch.distributed.broadcast) -- this replaces the NCCL parameter
 * sync of nn.DataParallelTable (util.lua:27-48, train.lua:494-496).  Call
 * b2f_commit_weights afterwards to rebuild the kernel-side packed copies.          */
B2F_API int b2f_weights_device(b2f_ctx *ctx, void **dev_ptr, long long *n);
B2F_API int b2f_commit_weights(b2f_ctx *ctx);
/* Host-only .t7 reader (replaces torch.load + nngraph walk): fills out[n] in canonical
 * order and sets *past_flow.  No GPU needed.                                       */
B2F_API int b2f_load_t7(const char *path, float *out, long long cap, long long *n, int *past_flow);
/* The same for any graph shape (torch.load of a model built by createModelMulti(opt), pwc.lua:88-121): graph_opts NULL / "" =
 * read win / levels / skip from the file, else the file must be that graph; opts_out (optional, opts_cap bytes) receives the
 * graph as an option string ("win=5,levels=4,skip=2,...,past_flow=1"), out[n] the weights in that graph's canonical order. */
B2F_API int b2f_load_t7_ex(const char *path, const char *graph_opts, float *out, long long cap, long long *n, char *opts_out,
                   int opts_cap);

/* ---- the hot path, host boundary: computeFlow (back2future.lua:47-95) ----
 * im1..im3: 3 x H0 x W0 planar RGB floats in [0,1] (what image.load returns).
 * flow: 2 x H0 x W0 doubles (raw network flow, channel 0 = x, rescaled to H0 x W0
 * exactly as :80-84);  fwd_occ / bwd_occ: H0 x W0 bytes (0/1), thresholds of
 * est[3][2] / est[3][1] at 0.6666 (:87-91).                                        */
B2F_API int b2f_compute_flow(b2f_ctx *ctx, const float *im1, const float *im2, const float *im3,
                     int H0, int W0, double *flow, unsigned char *fwd_occ,
                     unsigned char *bwd_occ);
/* Same on n independent triplets (host buffers, n x 3 x H0 x W0 each frame set; outputs
 * n x 2 x H0 x W0 and n x H0 x W0).  The library overlaps uploads, kernels and downloads
 * of consecutive sub-batches; planes whose values are all k/255 cross the link as bytes
 * (rebuilt bit for bit on the device), the flow as fp32 widened on the host.          */
B2F_API int b2f_compute_flow_batch(b2f_ctx *ctx, int n, const float *im1, const float *im2,
                           const float *im3, int H0, int W0, double *flow,
                           unsigned char *fwd_occ, unsigned char *bwd_occ);
/* Same for frames that are still 8-bit (n x 3 x H0 x W0 bytes, planar RGB): the value of a
 * sample is byte / 255, the float image.load() [torch/image] makes of it, so the results are
 * those of b2f_compute_flow_batch on the converted floats, bit for bit, at a quarter of the
 * upload ("n x 9 x H x W f32 or ... u8").                                     */
B2F_API int b2f_compute_flow_batch_u8(b2f_ctx *ctx, int n, const unsigned char *im1,
                              const unsigned char *im2, const unsigned char *im3, int H0, int W0,
                              double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ);

/* ---- one node, several GPUs: replaces nn.DataParallelTable (util.lua:27-48) for this path ----
 * One context per GPU inside the calling process, one library worker thread per GPU (the
 * reference's replicas each run on their own thread, util.lua:34-40).  b2f_init_multi loads /
 * generates the weights once (first device) and broadcasts them into the other replicas' device
 * buffers with RCCL (ncclBroadcast, xGMI inside a node; dlopen'ed) or hipMemcpyPeer when RCCL is
 * not loadable -- the NCCL parameter sync of train.lua:494-496.  devices: n_gpus ordinals or NULL
 * (0 .. n_gpus-1); n_gpus = 0 takes every visible GPU.  The batch entry points split the n triplets
 * statically and contiguously over the GPUs (b2f_shard_range: the dim-1 split of util.lua:32;
 * sizes differ by at most one) and every GPU writes straight into its slice of the caller's
 * buffers; same arguments and results as b2f_compute_flow_batch[_u8].                          */
typedef struct b2f_multi b2f_multi;
B2F_API int b2f_init_multi(const char *name_or_path, int n_gpus, const int *devices, b2f_multi **out);
B2F_API void b2f_destroy_multi(b2f_multi *m);
/* transport: 0 one GPU, 1 RCCL broadcast, 2 hipMemcpyPeer */
B2F_API int b2f_multi_info(const b2f_multi *m, int *n_gpus, int *devices, int cap, int *transport);
/* the i-th replica (for b2f_info / b2f_set_option / b2f_set_weights on replica 0 followed by
 * b2f_multi_rebroadcast); NULL if out of range.  Owned by m.                                    */
B2F_API b2f_ctx *b2f_multi_context(b2f_multi *m, int i);
B2F_API int b2f_multi_rebroadcast(b2f_multi *m);
/* FNV-1a of every replica's weight buffer as it sits on its GPU: equal after a broadcast        */
B2F_API int b2f_multi_weights_checksum(b2f_multi *m, unsigned long long *sums, int cap);
/* host-only: [lo, hi) of `rank` when n items are split over `world` GPUs                         */
B2F_API int b2f_shard_range(int n, int rank, int world, int *lo, int *hi);
B2F_API int b2f_multi_compute_flow_batch(b2f_multi *m, int n, const float *im1, const float *im2,
                                 const float *im3, int H0, int W0, double *flow,
                                 unsigned char *fwd_occ, unsigned char *bwd_occ);
B2F_API int b2f_multi_compute_flow_batch_u8(b2f_multi *m, int n, const unsigned char *im1,
                                    const unsigned char *im2, const unsigned char *im3, int H0, int W0,
                                    double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ);

/* ---- the hot path, device boundary: model:forward(imgs) (back2future.lua:74) ----
 * dev_in: B x 9 x H x W planar fp32 on the GPU (the tensor `imgs` of :73), H and W
 * multiples of 64.  in_kind: B2F_IN_NORMALIZED = already colour-normalized (what the
 * reference feeds the model), B2F_IN_UNIT = raw [0,1] values, normalized on device.
 * Outputs (any may be NULL), all planar fp32 on the GPU:
 *   dev_flow  B x 2 x H x W   est[1] = skip_ufs[3]
 *   dev_occ   B x 2 x H x W   skip_occs[3] (softmax probabilities)
 *   dev_est3  B x C3 x H x W  est[3] as computeFlow reads it: C3 = 2 (Soft: the
 *                             occlusion map) or 3 (Hard: warped image 1)
 * Every device pointer must be 16-byte aligned (vector loads / stores; hipMalloc and torch
 * allocations are, a view at an odd storage offset is not): a misaligned one is rejected.
 * The call is asynchronous on `stream`; the context's own stream (NULL) is a blocking stream
 * (ordered with the legacy default stream, not with other non-blocking streams); results are
 * awaited with b2f_synchronize.                                            */
enum { B2F_IN_NORMALIZED = 0, B2F_IN_UNIT = 1 };
B2F_API int b2f_forward_device(b2f_ctx *ctx, const void *dev_in, int in_kind, int B, int H, int W,
                       float *dev_flow, float *dev_occ, float *dev_est3, void *stream);
/* Full output table of model:forward (pwc.lua:459-489) into n_outs host buffers, in
 * table order; x is B x 9 x H x W normalized host memory.                           */
B2F_API int b2f_forward(b2f_ctx *ctx, const float *x, int B, int H, int W, float **outs, int n_outs);
B2F_API int b2f_output_shapes(const b2f_ctx *ctx, int H, int W, int *ch, int *oh, int *ow, int cap);

/* Execution options: use_graph (default 0) = b2f_forward_device replays a hipGraph per
 * (shape, pointers) combination, captured on its second use; host_graph (default 1) = the
 * same inside b2f_compute_flow*; profile = record HIP events around every kernel launch
 * (eager mode); profile_layers = one profile row per layer shape.
 * Kernel selection: wino4_min_pixels (default 4096) = stride-1 convs run the Winograd F(4x4)
 * kernel on maps of at least that many pixels and F(2x2) below -- a rule of the map size only,
 * so a triplet's result never depends on the batch it is computed in; wino_split_pixels (default
 * 512) = F(2x2) launches on maps of at most that many pixels run one block per 32 outputs
 * (twice the blocks for the coarsest level; same bits); adaptive_kernels = 1
 * picks the variant per launch by block rounds on the 256 CUs instead (faster for single
 * triplets, results then vary at the 1e-6 level with the batch size); corr_variant (-1 auto /
 * 0 .. 7) forces an instantiation of the warp + cost-volume kernel (same bits either way);
 * wino4_persistent (default 1) = F(4x4) launches run as persistent blocks, one per CU
 * (0: one tile per block; same bits either way); s2_tiles_per_block
 * (default 0 = launcher's rule) = tiles a block of the stride-2 kernel chains (same bits);
 * wino8 / s2_tile_groups (default 1) = launches that cannot fill the chip (a single triplet's
 * coarse levels) run the eight-wave F(2x2) kernel / one 32-output tile per stride-2 block
 * (more, lighter blocks; same bits).
 * Host pipeline of b2f_compute_flow*: host_subbatch_pixels, host_threads (0 = auto), host_u8,
 * host_ramp.  The library reads no environment variable after b2f_init (which takes
 * B2F_<OPTION> as the initial value of the tuning options).                              */
B2F_API int b2f_set_option(b2f_ctx *ctx, const char *key, int value);
B2F_API int b2f_get_option(const b2f_ctx *ctx, const char *key, int *value);
/* Per-kernel-class timings gathered while profile=1.  names: cap x 32 chars.        */
B2F_API int b2f_profile_read(b2f_ctx *ctx, char *names, double *total_ms, long long *launches,
                     int cap, int *n);
B2F_API int b2f_profile_reset(b2f_ctx *ctx);
B2F_API int b2f_synchronize(b2f_ctx *ctx);

/* ---- op-level entry points (host pointers, layouts of the reference modules) ----
 * nn.CostVolMulti(win, fwd):updateOutput({ref, frm}) -- models/CostVolMulti.lua:49-109;
 * ref, frm: B x C x h x w; out: B x win*win x h x w.                               */
B2F_API int b2f_op_costvol(b2f_ctx *ctx, const float *ref, const float *frm, int B, int C, int h,
                   int w, int win, int fwd, float *out);
/* nn.BilinearSamplerBHWD:updateOutput({img, grid}), CUDA semantics --
 * extras/stnbhwd/BilinearSamplerBHWD.cu:41-158; img: B x ih x iw x C, grid:
 * B x gh x gw x 2 (x first), out: B x gh x gw x C.                                  */
B2F_API int b2f_op_warp_bhwd(b2f_ctx *ctx, const float *img, const float *grid, int B, int ih,
                     int iw, int C, int gh, int gw, float *out);
/* Backward passes of the two custom modules (training side; not used by computeFlow).
 * nn.BilinearSamplerBHWD:updateGradInput -- BilinearSamplerBHWD.lua:81-107, CUDA kernel
 * BilinearSamplerBHWD.cu:161-307: grad_out B x gh x gw x C -> grad_img B x ih x iw x C (zeroed, then
 * accumulated; NULL = the onlyGrid instantiation, :372-421) and grad_grid B x gh x gw x 2 (x first).      */
B2F_API int b2f_op_warp_bhwd_backward(b2f_ctx *ctx, const float *img, const float *grid,
                              const float *grad_out, int B, int ih, int iw, int C, int gh, int gw,
                              float *grad_img, float *grad_grid);
/* nn.CostVolMulti(win, fwd):updateGradInput({ref, frm}, grad_out) -- models/CostVolMulti.lua:111-181:
 * grad_out B x win*win x h x w -> grad_ref, grad_frm B x C x h x w.                                       */
B2F_API int b2f_op_costvol_backward(b2f_ctx *ctx, const float *ref, const float *frm,
                            const float *grad_out, int B, int C, int h, int w, int win, int fwd,
                            float *grad_ref, float *grad_frm);
/* The fused kernel the pipeline uses for pwc.lua:246-267 + :393-409: warp both
 * neighbour maps by +k*flow (future) / -k*flow (past) and emit the joined
 * 162-channel cost volume.  ref/nbr_future/nbr_past: B x C x h x w; flow: B x 2 x h x w
 * or NULL (level 7: no warp); out: B x 162 x h x w.                                 */
B2F_API int b2f_op_warp_costvol(b2f_ctx *ctx, const float *ref, const float *nbr_future,
                        const float *nbr_past, const float *flow, float k, int B, int C,
                        int h, int w, float *out);
/* nn.SpatialConvolution(Ci,Co,3,3,s,s,1,1) [+ nn.LeakyReLU(0.2)] -- pwc.lua:58-85;
 * x: B x Ci x H x W, w: Co x Ci x 3 x 3, y: B x Co x Ho x Wo.                       */
B2F_API int b2f_op_conv3x3(b2f_ctx *ctx, const float *x, int B, int Ci, int H, int W, const float *w,
                   const float *bias, int Co, int stride, int leaky, float *y);
/* The two 16-channel layers of the head of the pyramid as the pipeline runs them with option bf16_direct = 2, in one kernel:
 * nn.SpatialConvolution(16,16,3,3,1,1,1,1) + LeakyReLU(0.2) (pwc.lua:62, level-2 convUnit) followed by
 * nn.SpatialConvolution(16,32,3,3,2,2,1,1) + LeakyReLU(0.2) (pwc.lua:60, level-3 convUnit).
 * x: B x 16 x H x W, w1: 16 x 16 x 3 x 3, w2: 32 x 16 x 3 x 3, y: B x 32 x ceil(H/2) x ceil(W/2).                       */
B2F_API int b2f_op_conv_head16(b2f_ctx *ctx, const float *x, int B, int H, int W, const float *w1, const float *b1,
                       const float *w2, const float *b2, float *y);
/* nn.SpatialUpSamplingBilinear(2) on a 2-channel flow field -- pwc.lua:360-381;
 * x: B x 2 x h x w -> y: B x 2 x 2h x 2w.                                            */
B2F_API int b2f_op_upsample_flow2x(b2f_ctx *ctx, const float *x, int B, int h, int w, float *y);
/* image.scale(src, Wd, Hd) 'bilinear' [torch/image] as computeFlow uses it -- back2future.lua:71 -- with
 * ColorNormalize (transforms.lua:33-45, plane % 3 = colour) applied first when normalize != 0;
 * src: C x Hs x Ws -> dst: C x Hd x Wd.  Bit-identical to the CPU routine.             */
B2F_API int b2f_op_image_scale(b2f_ctx *ctx, const float *src, int C, int Hs, int Ws, int normalize,
                       float *dst, int Hd, int Wd);

#ifdef __cplusplus
}
#endif
#endif /* B2F_H */
