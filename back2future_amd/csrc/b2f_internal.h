// Internal C++ interface between the C-ABI layer (b2f_api.hip) and the gfx950
// kernels (b2f_conv.hip, b2f_corr.hip, b2f_glue.hip).  Device layout everywhere:
// NHWC ("BHWD") fp32; see DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
// B2F_EXPERIMENTS=1 (python -m back2future_amd.build --experiments -> libb2f_exp.so): the kernels of tools/experiments/csrc -- forms that
// were built, tested and measured no faster than the defaults (profiles/r04_*_notes.txt) -- and the options that select them.  The
// product library is built without them.
#ifndef B2F_EXPERIMENTS
#define B2F_EXPERIMENTS 0
#endif
#include <cstdint>

namespace b2f {

// Architecture constants of the shipped graph (models/pwc.lua:29,76-89 with
// opts.lua:83-98).
constexpr int kLevels = 7, kLst = 3, kWin = 9, kND = 2 * kWin * kWin;  // 162
constexpr int kFeat[8] = {0, 3, 16, 32, 64, 96, 128, 192};
constexpr int kDec[7] = {0, 128, 128, 96, 64, 32, 2};
// Activation layout on the device: "chunk-planar" fp32, [image][C/8][h][w][8] -- planes of
// 8-channel chunks.  An 8-channel chunk is the K step of the conv kernel and of the cost-volume
// kernel, and in this layout the chunk of neighbouring pixels is contiguous (32 B per pixel, four
// pixels per 128-B line), so halo tiles and bilinear gathers touch whole cache lines instead of
// 32 B out of every pixel's NHWC record.  Kernels take (img_stride, chunk_stride, pix_stride) in
// floats, so NHWC is still expressible (chunk_stride = 8, pix_stride = C) for the op-level API.
//
// One cost-volume pixel record = 21 chunks = 168 floats, slot order
//   [fwd 0..79 | bwd 0..79 | fwd80, bwd80, u, v, ub, vb, 0, 0]
// (u,v = ufs[l+1], ub,vb = ubfs[l+1]: the upsampled flows the decoders also take, pwc.lua:334,337).
// Each direction's first 80 channels fill ten whole chunks, so a thread stores float4 pairs.
constexpr int kCvRec = 168;
constexpr int kCvChunks = 21;
constexpr int kImgC = 8;   // packed image channels (RGB + zero pad), one conv K-chunk
constexpr int kCK = 8;     // conv K-chunk: input channels staged per LDS pass

// ---- conv3x3 (MFMA implicit GEMM) -------------------------------------------------
struct ConvSeg {
    const float *ptr;
    long img_stride;    // floats between consecutive images
    long chunk_stride;  // floats between consecutive 8-channel chunks of one pixel
    int pix_stride;     // floats between consecutive pixels
    int nchunks;        // K-chunks (of kCK channels) taken from this segment
};
struct ConvLaunch {
    ConvSeg seg[2];
    int nseg;
    const float *wpk;   // packed weights [nblk][chunk][9][2][NT*32][4]
    const float *bias;  // [nblk*NT*32]
    float *out;
    long out_img_stride;
    long out_chunk_stride;
    int out_pix_stride;
    int cout;           // valid output channels
    int nt, nblk;       // 32-wide N tiles per block, N blocks
    int H, W, Ho, Wo, stride;
    int nimg;
    int leaky;
    long long *trace;   // profiling builds only (B2F_WINO_TRACE), nullptr otherwise
    int nb0;            // first n-block of this launch (Winograd kernel: a layer may be split over two launches)
    int nsplit;         // F(2x2) kernel: 1 = one block per 32-output N tile of the 64-wide packing (more, lighter blocks for small launches)
    int w8 = 1;                // F(2x2) kernel: one-N-tile launches of at most one block per CU run the eight-wave form (same bits)
    int w4_persist = 1;        // F(4x4) kernel: persistent blocks (one per CU) when the launch has at least two tiles per CU
    int tiles_per_block = 0;   // direct kernel: consecutive tiles a block chains into one (tile, chunk) pipeline; 0 = launcher's choice
    int w4_hybrid = 0;                 // F(4x4) persistent two-N-tile kernel: xi steps per wave that run on the bf16 pipe with split operands (0 = none)
    const void *wpk_split2 = nullptr;  // F(2x2) kernel on the bf16 pipe (b2f_wino2s.hip): its split weights, or null
    const void *wpk_bf6 = nullptr;     // direct kernel on the bf16 pipe (b2f_convb.hip): weights pre-split into bf16 windows, or null
    const float *bias_bf6 = nullptr;   // ... and its bias, padded to blocks of 64 outputs
    int bf16_direct = 0;               // direct 16-channel kernels on the bf16 pipe with split fp32 operands (b2f_conv16b.hip)
    const void *wpk_split = nullptr;   // F(4x4) kernel on the bf16 pipe (b2f_wino4s.hip): weights split into three bf16 terms, or null
    const void *wpk_w1b = nullptr;     // 1-D Winograd F(4,3) kernel on the bf16 pipe (b2f_w1b.hip): its split weights, or null
    const float *bias_w1b = nullptr;   // ... and its bias, padded to blocks of 64 outputs
    const void *wpk_s2b = nullptr;     // stride-2 loader / consumer kernel on the bf16 pipe (b2f_s2b.hip): its split weights, or null
    const float *bias_s2b = nullptr;   // ... and its bias, padded to tiles of 32 outputs
    int w1b_nblk = 0;                  // ... n-blocks of 64 outputs it computes (the first ones; 0 = all)
    const void *wpk_w6 = nullptr;      // Winograd F(6x6) kernel (b2f_wino6.hip): its weights, or null
    const float *bias_w6 = nullptr;    // ... and its bias, padded to blocks of 64 outputs
};
hipError_t launch_conv3x3(const ConvLaunch &p, hipStream_t s);
// floats needed for the packed weights of a conv with `cin_chunks` K-chunks
size_t conv_wpk_floats(int cin_chunks, int nt, int nblk);
void conv_choose_tiles(int cout, int *nt, int *nblk);
// Host-side re-pack Co x Ci x 3 x 3 (Torch) -> kernel layout.  cin_map[k] gives, for
// packed input channel k (0 .. chunks*8-1), the Torch input channel or -1 (zero).
void conv_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map,
                       int cin_chunks, int nt, int nblk, float *wpk, float *bpk);

// ---- 16 -> 16 stride-1 layer (level-2 convUnit): single-pass kernel on the 16x16x4 MFMA (b2f_conv16.hip),
// weights [tap 9][kg 4][co 16][4]
hipError_t launch_conv3x3_c16(const ConvLaunch &p, hipStream_t s);
hipError_t launch_conv3x3_c16b(const ConvLaunch &p, hipStream_t s);   // the same layer on the bf16 pipe (ConvLaunch::bf16_direct)
size_t c16_wpk_floats();
void c16_pack_weights(const float *w, const float *b, int Ci, const int *cin_map, float *wpk, float *bpk);
// direct conv (stride 1 / 2) as an implicit GEMM on the bf16 pipe with split fp32 operands (b2f_convb.hip); weights
// [n-block of 64][chunk][tap 9][k4 2][co 64] x 32 bytes
bool convb_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_bf6(const ConvLaunch &p, hipStream_t s);
int convb_nblk(int cout);
size_t convb_wpk_floats(int cin_chunks, int cout);
void convb_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk);
// stride-2 layers as a direct implicit GEMM on the bf16 pipe with split fp32 operands in loader / consumer persistent blocks that compute all
// outputs of a tile (b2f_s2b.hip); weights [chunk][tap 9][window 2][output tile][kh 2][co 32] x 16 bytes
bool s2b_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_s2b(const ConvLaunch &p, hipStream_t s);
int s2b_ntiles(int cout);
size_t s2b_wpk_floats(int cin_chunks, int cout);
void s2b_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk);
// stride-1 layers as a one-dimensional Winograd F(4,3) along x on the bf16 pipe with split fp32 operands, loader / consumer
// persistent blocks (b2f_w1b.hip); weights [n-block of 64][chunk][step 18][window 2][N tile 2][kh 2][co 32] x 16 bytes
bool w1b_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_w1b(const ConvLaunch &p, hipStream_t s);
int w1b_nblk(int cout);
size_t w1b_wpk_floats(int cin_chunks, int cout);
void w1b_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk);
// stride-1 layers as Winograd F(6x6,3x3) on the fp32 MFMA (b2f_wino6.hip): blocks of 16 tiles x 64 outputs, transformed input in registers;
// weights [n-block of 64][chunk][row a 8][step 16][lane 64][4] (a last block of <= 32 outputs: half of that)
bool wino6_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_wino6(const ConvLaunch &p, hipStream_t s);   // blocks of 64 outputs, then one of 32 when the outputs are <= 32 mod 64
int wino6_nblk(int cout);
size_t wino6_wpk_floats(int cin_chunks, int cout);
void wino6_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk);
// 16 -> 16 (stride 1) + 16 -> 32 (stride 2), both with LeakyReLU(0.2), as ONE streaming kernel on the bf16 pipe (b2f_head.hip): the
// 16-channel map between them stays in LDS.  Weights: the c16 / c16s2 packings.
struct HeadLaunch {
    const float *in;                    // 16 channels, chunk-planar
    long in_img_stride, in_chunk_stride;
    int in_pix_stride;
    int H1, W1;                         // size of the input = size of the intermediate
    const float *w1, *b1, *w2, *b2;
    float *out;                         // 32 channels at Ho x Wo
    long out_img_stride, out_chunk_stride;
    int out_pix_stride;
    int Ho, Wo, nimg;
    int rows_per_block, nsy, nsx;       // filled in by the launcher
    long long *trace = nullptr;         // profiling builds only
};
bool head16_supported(const HeadLaunch &p);
hipError_t launch_conv_head16(HeadLaunch p, hipStream_t s);
// 16 -> 32 channels, stride 2 (mode 5): the same single-pass scheme
bool c16s2_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_c16s2(const ConvLaunch &p, hipStream_t s);
size_t c16s2_wpk_floats();
void c16s2_pack_weights(const float *w, const float *b, int Ci, const int *cin_map, float *wpk, float *bpk);
// ---- 2-output stride-1 layers (last decoder layer): VALU kernel in b2f_glue.hip, weights [chunk][tap][8][2]
hipError_t launch_conv_narrow2(const ConvLaunch &p, hipStream_t s);
size_t narrow2_wpk_floats(int cin_chunks);
void narrow2_pack_weights(const float *w, const float *b, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk);
// ---- Winograd F(2x2,3x3) variant for stride-1 layers (b2f_wino.hip); same ConvLaunch, weights
// packed by wino_pack_weights ([nblk][chunk][xi 16][k4 2][NT*32][4]), nt in {1, 2}
hipError_t launch_conv3x3_wino(const ConvLaunch &p, hipStream_t s);
void wino_choose_tiles(int cout, int *nt, int *nblk);
size_t wino_wpk_floats(int cin_chunks, int nt, int nblk);
// ---- Winograd F(4x4,3x3) variant for the wide stride-1 layers (b2f_wino4.hip): 64 output channels per
// n-block, weights packed by wino4_pack_weights ([nblk][chunk][xi 36][k4 2][64][4])
hipError_t launch_conv3x3_wino4(const ConvLaunch &p, hipStream_t s);
hipError_t launch_conv3x3_wino4_rem(const ConvLaunch &p, hipStream_t s);   // its last, half-empty n-block only
// the same layers on the bf16 matrix pipe with exactly split fp32 operands (b2f_wino4s.hip): n-blocks [nb0, nb0 + nblk) of
// 64 outputs each, persistent blocks; weights packed by wino4s_pack_weights
bool wino4s_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_wino4s(const ConvLaunch &p, int nb0, int nblk, hipStream_t s);
size_t wino4s_wpk_floats(int cin_chunks, int nblk);
// Winograd F(2x2) on the bf16 matrix pipe with split operands (b2f_wino2s.hip): 64 tiles x 64 outputs per block
bool wino2s_supported(const ConvLaunch &p);
hipError_t launch_conv3x3_wino2s(const ConvLaunch &p, int nb0, int nblk, hipStream_t s);
size_t wino2s_wpk_floats(int cin_chunks, int nblk);
void wino2s_pack_weights(const float *w, int Co, int Ci, const int *cin_map, int cin_chunks, int nblk, float *wpk);
void wino4s_pack_weights(const float *w, int Co, int Ci, const int *cin_map, int cin_chunks, int nblk, float *wpk);
int wino4_nblk(int cout);
size_t wino4_wpk_floats(int cin_chunks, int nblk);
void wino4_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks,
                        int nblk, float *wpk, float *bpk);
void wino_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map,
                       int cin_chunks, int nt, int nblk, float *wpk, float *bpk);

// ColorNormalize's (x + (-mean_c)) / std_c (transforms.lua:33-45; constants of back2future.lua:33-36) with the IEEE quotient but
// without the ~11-instruction float division: y = RN(1 / std), a' = a / 16, q = a' y, r = fma(-q, std, a'), q' = fma(r, y, q),
// result 16 q' (Markstein's correction step on a scaled dividend: q cannot overflow before the quotient does; a itself where a
// is inf / NaN).  tools/div_const_check.hip compares it with __fdiv_rn on the GPU for EVERY float32 bit pattern of x and the
// three channels: 0 of 3 x 2^32 results differ.
__device__ __forceinline__ float color_normalize(float x, int c)
{
    constexpr float mean[3] = {0.485f, 0.456f, 0.406f};
    constexpr float stdv[3] = {0.229f, 0.224f, 0.225f};
    constexpr float rcp[3] = {(float)(1.0 / (double)0.229f), (float)(1.0 / (double)0.224f), (float)(1.0 / (double)0.225f)};
    const float a = x + (-mean[c]);              // add(-mean) then div(std)
    const float as = a * 0.0625f;
    const float q = as * rcp[c];
    const float r = __builtin_fmaf(-q, stdv[c], as);
    float f = __builtin_fmaf(r, rcp[c], q) * 16.f;
    if (!(__builtin_fabsf(a) < __builtin_inff())) f = a;
    return f;
}

// ---- fused warp + cost volume -----------------------------------------------------
struct CorrLaunch {
    const float *ref, *nbr_fut, *nbr_past;  // C channels each
    long img_stride, chunk_stride;           // of the three maps
    int pix_stride;
    const float *flow;                       // B x h x w x 2 (ufs) or nullptr
    const float *flow_b;                     // past flow ubfs (only copied into the record) or nullptr
    float k;                                 // warp scale of the future frame (+k), past = -k
    float *out;                              // records, kCvRec floats per pixel
    long out_img_stride, out_chunk_stride;
    int out_pix_stride;
    int B, C, h, w;
    int ablate = 0;                          // profiling only (option corr_ablate): 1 no gather loads, 2 no FMAs, 4 no stores, 8 no XCD remap
    int variant = -1;                        // -1 auto, 0 regular, 1 latency, 2 two-pixel, 3 two-pixel one-direction-per-block instantiation (same bits)
};
// hipFuncSetAttribute is per device: launchers keep one "done" flag per device (b2f_init_multi drives several GPUs from one
// process, one worker thread each); returns the slot of the current device
inline int attr_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev & 63;
}
// compute units of the current device (cached per device; 256 on an MI355X in SPX mode, fewer in a partition mode): the launchers'
// "does this launch fit in one round of blocks" thresholds scale with it
inline int device_cu_count()
{
    static int n_cu_dev[64] = {0};
    int &n = n_cu_dev[attr_slot()];
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v < 8 ? 8 : v;
    }
    return n;
}
#ifdef __HIPCC__
// MI355X dispatches consecutive workgroup ids round-robin over its 8 XCDs (8 private L2s).  Give every XCD one
// contiguous range of logical work items instead (bijective for any grid size; placement only affects speed).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
// getTopLeft of the CUDA sampler (BilinearSamplerBHWD.cu:6-20): the coordinate is clamped to the border FIRST, then
// floored; wt = weight of the top / left tap.  The one definition every warp in the library uses.
__device__ __forceinline__ void bhwd_top_left(float coord, int size, int &pt, float &wt)
{
    float c = coord;
    if (c < 0.f) c = 0.f;
    if (c > (float)(size - 1)) c = (float)(size - 1);
    const float fl = floorf(c);
    pt = (int)fl;
    wt = 1.f - (c - fl);
}
#endif

// slot of cost-volume channel c (0..80) of direction dir (0 fwd, 1 bwd) inside a record
inline int cv_slot(int dir, int c) { return c < 80 ? dir * 80 + c : 160 + dir; }
hipError_t launch_warp_costvol(const CorrLaunch &p, hipStream_t s);
// b2f_corr5.hip: the persistent "unit" form (variant 5; C a multiple of 32), same bits as the others
bool warp_costvol_unit_supported(const CorrLaunch &p);
hipError_t launch_warp_costvol_unit(const CorrLaunch &p, hipStream_t s);
// the role-specialised form of it (variant 6): FMA waves and gather waves, source window by LDS-DMA
hipError_t launch_warp_costvol_spec(const CorrLaunch &p, hipStream_t s);
// ten unit waves + six gather waves per block (variant 7)
hipError_t launch_warp_costvol_gw(const CorrLaunch &p, hipStream_t s);
// generic (any odd win) single-direction cost volume, NHWC in, B x h x w x win*win out
hipError_t launch_costvol_generic(const float *ref, const float *frm, int B, int C, int h, int w,
                                  int win, int fwd, float *out, hipStream_t s);

// ---- glue kernels -------------------------------------------------------------------
// planar B x 9 x H x W -> 3 frame-major NHWC8 images ([3][B][H][W][8]), optional normalize
hipError_t launch_pack_input(const float *in, int normalize, int B, int H, int W, float *img,
                             hipStream_t s);
// first pyramid layer (ColorNormalize + Narrow + conv 3->16 s2 + LeakyReLU) from the planar input;
// wt = [27 taps (c,ky,kx)][16 cout], out = chunk-planar [3][B][2][H/2*W/2][8]
hipError_t launch_conv_first(const float *in, int normalize, int B, int H, int W, const float *wt,
                             const float *bias, float *out, hipStream_t s);
// warp frame `frame` of the planar input (normalized on the fly) by k * planar flow
hipError_t launch_warp_input_planar(const float *in, int normalize, int frame, const float *flow_planar,
                                    float k, int B, int H, int W, float *out, hipStream_t s);
// nn.BilinearSamplerBHWD forward (CUDA semantics), grid scaled by k
hipError_t launch_warp_nhwc(const float *img, long img_stride, int pix_stride, int C, int ih,
                            int iw, const float *grid, float k, int B, int gh, int gw,
                            float *out, int out_pix_stride, hipStream_t s);
// SpatialUpSamplingBilinear(2) on a 2-channel field stored with `in_pix_stride` floats per pixel
// (2 = packed, 8 = first two channels of a chunk) -> packed B x 2h x 2w x 2
hipError_t launch_upsample_flow2x(const float *in, int in_pix_stride, int B, int h, int w, float *out,
                                  hipStream_t s);
// second x2 + packed -> planar B x 2 x 2h x 2w
hipError_t launch_upsample_flow2x_planar(const float *in, int in_pix_stride, int B, int h, int w,
                                         float *out, hipStream_t s);
// softmax over 2 logits (first two floats of each `in_pix_stride` record) + nearest x4 -> planar
hipError_t launch_softmax_nearest4_planar(const float *logits, int in_pix_stride, int B, int h, int w,
                                          float *out, hipStream_t s);
// warp full-res image 1 (packed NHWC8, RGB in ch 0..2) by k * planar flow -> planar B x 3 x H x W
hipError_t launch_warp_image_planar(const float *img8, const float *flow_planar, float k, int B,
                                    int H, int W, float *out, hipStream_t s);
// 2x2 average pooling on NHWC (C multiple of 4)
hipError_t launch_avgpool2_nhwc(const float *in, int nimg, int H, int W, int C, float *out,
                                hipStream_t s);
// NHWC (pix_stride, first C channels) -> planar B x C x h x w
hipError_t launch_nhwc_to_planar(const float *in, int pix_stride, int C, int B, int h, int w,
                                 float *out, hipStream_t s);
// chunk-planar [B][C/8][h][w][8] -> planar B x C x h x w
hipError_t launch_cp8_to_planar(const float *in, int C, int B, int h, int w, float *out, hipStream_t s);
hipError_t launch_planar_to_nhwc(const float *in, int C, int B, int h, int w, float *out,
                                 int pix_stride, hipStream_t s);
hipError_t launch_fill(float *p, size_t n, float v, hipStream_t s);

// ---- computeFlow boundary on the device (b2f_boundary.hip) ----------------------------------
// image.scale 'bilinear' of `planes` planes Hs x Ws -> Hd x Wd (tmp: planes x Hs x Wd floats), optionally with
// ColorNormalize applied to the source samples (plane % 3 = colour); bit-identical to the host function
hipError_t launch_image_scale(const float *src, int normalize, long planes, int Hs, int Ws, float *tmp, float *dst,
                              int Hd, int Wd, hipStream_t s);
// nearest rescale + thresholds (back2future.lua:77-93): planar net outputs -> fp32 flow at H0 x W0 (nullptr: not
// written, the net size is the output size), u8 masks
hipError_t launch_postprocess(const float *flow_net, const float *est3, int est3_ch, int B, int fh, int fw, int H0, int W0,
                              float *flow32, unsigned char *fwd_occ, unsigned char *bwd_occ, hipStream_t s);
// out[i] = in[i] / 255 (correctly rounded): device end of the 8-bit input transport
hipError_t launch_unpack_u8(const unsigned char *in, size_t n, float *out, hipStream_t s);

}  // namespace b2f
