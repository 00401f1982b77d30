// Internal to libb2f.so: the context behind the opaque b2f_ctx of include/b2f.h, shared by the C-ABI layer
// (b2f_api.hip) and the host-buffer pipeline (b2f_pipeline.hip).
#pragma once
#include "../../include/b2f.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "b2f_host.h"
#include "b2f_internal.h"

namespace b2f {

// error reporting of the C ABI: message kept per thread for b2f_last_error(), returns 1
int api_fail(const std::string &m);
const std::string &api_error();


struct PackedConv {
    int nseg = 1;
    int chunks[2] = {0, 0};
    int cout = 0, nt = 1, nblk = 1;
    int wino = 0;                  // 0 direct kernel, 1 VALU kernel for 2 outputs, 2 Winograd F(2x2,3x3), 3 16->16 kernel,
                                   // 4 Winograd F(4x4,3x3)
    size_t w_off = 0, b_off = 0;   // float offsets into wpk_dev
    // F(4x4) layers also carry an F(2x2) packing for small maps (one 16 x 32-pixel block per CU does not fill
    // the chip below ~64 x 64 pixels; measured at 32 x 60: 0.05 ms vs 0.14 ms)
    int nt2 = 0, nblk2 = 0;
    size_t w_off2 = 0, b_off2 = 0;
    // ... and the weights split into three bf16 terms for the kernel on the bf16 matrix pipe (b2f_wino4s.hip); 0 = none
    size_t w_off3 = 0;
    size_t w_off4 = 0;             // F(2x2) split packing (b2f_wino2s.hip); 0 = none
    size_t w_off5 = 0, b_off5 = 0; // direct layers: weights pre-split for the bf16-pipe kernel (b2f_convb.hip) + bias padded to 64; 0 = none
    size_t w_off7 = 0, b_off7 = 0; // stride-2 layers: packing of the loader / consumer kernel (b2f_s2b.hip); 0 = none
    size_t w_off8 = 0, b_off8 = 0; // F(4x4)-class layers: Winograd F(6x6) packing (b2f_wino6.hip); 0 = none
    size_t w_off6 = 0, b_off6 = 0; // F(4x4)-class layers: 1-D Winograd F(4,3) packing for the bf16-pipe loader / consumer kernel (b2f_w1b.hip); 0 = none
};

struct ProfEvent {
    int name;
    hipEvent_t a, b;
};

struct GraphKey {
    const void *in;
    float *flow, *occ, *est3;
    int kind, B, H, W;
    int rule;                      // kernel-choice rule the captured launches follow (1 = per launch: a single-triplet request), part of the key
    bool operator<(const GraphKey &o) const
    {
        return std::tie(in, flow, occ, est3, kind, B, H, W, rule) < std::tie(o.in, o.flow, o.occ, o.est3, o.kind, o.B, o.H, o.W, o.rule);
    }
};

// One of the two buffer sets the host-buffer entry point (b2f_compute_flow_batch) alternates between: while the
// kernels of sub-batch k run on set k & 1, the uploads of k + 1 and the downloads of k - 1 use the other one.
struct HostSlot {
    char *dev = nullptr;        // device blob, carved below
    size_t dev_bytes = 0;
    char *pin = nullptr;        // pinned staging blob
    size_t pin_bytes = 0;
    unsigned char *d_u8 = nullptr;   // 8-bit transport of the input planes (see pack_u8_piece)
    float *d_up = nullptr, *d_tmp = nullptr, *d_in = nullptr, *d_flow = nullptr, *d_est3 = nullptr;
    float *d_flow32 = nullptr;       // flow at H0 x W0 (fp32, before the f64 sc_w / sc_h factors); = d_flow without a rescale
    unsigned char *d_fo = nullptr, *d_bo = nullptr;
    unsigned char *h_u8 = nullptr;
    float *h_in = nullptr, *h_flow32 = nullptr;
    unsigned char *h_fo = nullptr, *h_bo = nullptr;
    hipEvent_t ev_in = nullptr, ev_comp = nullptr, ev_out = nullptr;
};

// Host work items of the pipeline, cut into pieces and spread over a pool of threads.
enum { JOB_COPY = 0, JOB_F32_TO_F64 = 1, JOB_PACK_U8 = 2 };
struct CopyJob {
    void *dst;
    const void *src;
    size_t bytes;                     // of the source
    int kind = JOB_COPY;
    double scale = 1.0;               // JOB_F32_TO_F64: dst = (double)src * scale   (back2future.lua:83-84)
    std::atomic<int> *inexact = nullptr;   // JOB_PACK_U8: set when a value is not k / 255
};

// 8-bit transport: image.load hands computeFlow floats that came from 8-bit files, i.e. k / 255.  Such a plane
// crosses the link as bytes (a quarter of the traffic of a path that is PCIe-bound) and is rebuilt on the device
// by the same correctly rounded division, but only if that reproduces every float of it bit for bit; one other
// value (or -0, NaN, ...) and the triplet is uploaded as floats instead.
inline bool pack_u8_piece(unsigned char *d, const float *s, size_t n)
{
    uint32_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const float v = s[i];
        int k = (int)(v * 255.0f + 0.5f);
        k = k < 0 ? 0 : (k > 255 ? 255 : k);
        const float r = (float)k / 255.0f;
        uint32_t vb, rb;
        memcpy(&vb, &v, 4);
        memcpy(&rb, &r, 4);
        bad |= vb ^ rb;
        d[i] = (unsigned char)k;
    }
    return bad == 0;
}

inline void run_piece(const CopyJob &j)
{
    if (j.kind == JOB_COPY) {
        memcpy(j.dst, j.src, j.bytes);
    } else if (j.kind == JOB_F32_TO_F64) {
        const float *s = (const float *)j.src;
        double *d = (double *)j.dst;
        const double sc = j.scale;
        for (size_t i = 0, n = j.bytes / 4; i < n; ++i) d[i] = (double)s[i] * sc;
    } else {
        if (!pack_u8_piece((unsigned char *)j.dst, (const float *)j.src, j.bytes / 4)) j.inexact->store(1, std::memory_order_relaxed);
    }
}

// Persistent host threads that execute job lists in 1 MB pieces (the caller's thread works too).  One core moves
// ~10 GB/s; a full-HD triplet is 71 MB in and 35 MB out, so single-threaded staging would cost several times the
// 1.5 ms the GPU needs for it.
class CopyPool {
public:
    explicit CopyPool(int workers)
    {
        for (int i = 0; i < workers; ++i) th_.emplace_back([this] { worker(); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    void run(const std::vector<CopyJob> &jobs)
    {
        constexpr size_t kPiece = 1 << 20;
        std::vector<CopyJob> pieces;
        size_t bytes = 0;
        for (const CopyJob &j : jobs) {
            const size_t dmul = j.kind == JOB_F32_TO_F64 ? 2 : 1, ddiv = j.kind == JOB_PACK_U8 ? 4 : 1;
            for (size_t o = 0; o < j.bytes; o += kPiece) {
                CopyJob q = j;
                q.dst = (char *)j.dst + o * dmul / ddiv;
                q.src = (const char *)j.src + o;
                q.bytes = std::min(kPiece, j.bytes - o);
                pieces.push_back(q);
            }
            bytes += j.bytes;
        }
        if (th_.empty() || bytes < (2u << 20)) {
            for (const CopyJob &j : pieces) run_piece(j);
            return;
        }
        std::unique_lock<std::mutex> l(m_);
        pieces_ = &pieces;
        next_ = done_ = 0;
        cv_.notify_all();
        while (next_ < pieces.size()) {
            const CopyJob j = pieces[next_++];
            l.unlock();
            run_piece(j);
            l.lock();
            ++done_;
        }
        cv_done_.wait(l, [&] { return done_ == pieces.size(); });
        pieces_ = nullptr;
    }

private:
    void worker()
    {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            cv_.wait(l, [&] { return stop_ || (pieces_ && next_ < pieces_->size()); });
            if (stop_) return;
            const CopyJob j = (*pieces_)[next_++];
            l.unlock();
            run_piece(j);
            l.lock();
            if (++done_ == pieces_->size()) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    const std::vector<CopyJob> *pieces_ = nullptr;
    size_t next_ = 0, done_ = 0;
    bool stop_ = false;
};

}  // namespace b2f

struct b2f_ctx {
    int device = 0;
    bool past_flow = false;
    b2f::GraphOpts g;           // graph shape (createModelMulti options); g.past_flow == past_flow
    long long nparams = 0;
    hipStream_t stream = nullptr;
    std::vector<b2f::ConvDesc> lay;
    std::vector<b2f::PackedConv> packed;
    float *w_dev = nullptr;     // flat canonical weights
    float *wpk_dev = nullptr;   // packed kernel-side copies
    size_t wpk_floats = 0;
    size_t first_w_off = 0, first_b_off = 0;   // [27][16] weights + bias of the first pyramid conv (conv_first_kernel)
    // workspace arena
    float *arena = nullptr;
    size_t arena_floats = 0;
    int wsB = 0, wsH = 0, wsW = 0;
    // options
    int use_graph = 0, profile = 0;
    int host_graph = 1;   // b2f_compute_flow*: replay hipGraphs for repeated (shape, sub-batch) combinations
    // kernel selection and pipeline tuning (b2f_set_option; seeded once from the environment in b2f_init, never read
    // from it on the hot path)
    int wino8 = 1;                 // F(2x2) one-N-tile launches of at most one block per CU run the eight-wave form (conv3x3_wino8; same bits)
    int wino_split_pixels = 512;   // F(2x2) launches on maps of at most this many pixels run one block per 32-output N tile (same bits;
                                   // level 7 of a full-HD triplet: 64 tiles per launch at batch 16 -- 0.25 -> 0.20 ms for its six layers)
    int wino4_min_pixels = 4096;   // F(4x4) for maps of at least this many pixels, F(2x2) below: depends on the map size
                                   // only, so a triplet's result does not depend on the batch it is computed in
    int cur_batch = 0;             // triplets of the REQUEST the forward pass being issued belongs to (run_conv's per-launch rule for single-triplet calls)
    int req_batch = 0;             // set by the host-buffer entry points to the caller's n while they issue their sub-batches (0: a forward call's own B):
                                   // the kernel choice follows the caller's request, so a triplet's bits do not depend on its position in a batch
    int adaptive_kernels = -1;     // -1 (default): per launch for single-triplet calls, by map size for batches; 0: by map size always;
                                   // 1: choose the Winograd variant per launch by block rounds on the 256 CUs (faster for
                                   // single triplets / small batches; results then depend on the batch size at 1e-6 level)
    int corr_ablate = 0;           // profiling only, see CorrLaunch::ablate
    int corr_variant = -1;         // warp + cost volume: -1 auto, 0 regular, 1 latency variant (bit-identical results)
    int op_wino_split = 0;         // b2f_op_conv3x3: F(2x2) kernel with one block per 32-output N tile (tests)
    int profile_layers = 0;        // one profile row per (layer shape, map size)
    int bf16_direct = 2;           // the 16-channel layers of the head on the bf16 matrix pipe with exactly split fp32 operands: 0 = fp32-MFMA kernels,
                                   // 1 = the 16 -> 16 layer alone (b2f_conv16b.hip), 2 = 16 -> 16 + 16 -> 32 stride 2 fused, the map between them in LDS (b2f_head.hip)
    int bf16_conv_min_pixels = 65536;   // bf16_conv = 2: 32-output stride-1 layers on maps of at least this many pixels leave the F(4x4) kernel
    int bf16_conv = 1;             // 1 (default): the direct (stride-2) layers on the bf16 matrix pipe with split fp32 operands (b2f_convb.hip);
                                   // 2: also the 32-output stride-1 layers of large maps, 3: every F(4x4)-class layer of large maps (experiments: measured
                                   // slower than the F(4x4) kernel, profiles/r04_bf16_direct_notes.txt (4)); 0: fp32-MFMA kernel
    int wino2_split = 0;           // F(4x4)-class layers, blocks of 64 outputs: 1 = Winograd F(2x2) on the bf16 matrix pipe with exactly split
                                   // fp32 operands (b2f_wino2s.hip) on maps of at least wino4_min_pixels pixels
    int wino4_hybrid = 0;          // F(4x4) two-N-tile blocks: this many of a wave's nine xi steps on the bf16 pipe with split operands (needs the
                                   // split packing: setting it > 0 packs it); 0 = all on the fp32 MFMA
    int wino4_split = 0;           // F(4x4) layers with two full N tiles per block: 1 = on the bf16 matrix pipe with exactly split fp32 operands
                                   // (b2f_wino4s.hip; fp32-level accuracy, measured no faster: profiles/r04_wino4s_notes.txt), 0 = on the fp32 MFMA
    int wino6 = 1;                 // F(4x4)-class layers on maps of at least wino6_min_pixels pixels: 1 = Winograd F(6x6,3x3) on the fp32 MFMA (csrc/b2f_wino6.hip):
                                   // blocks of 64 outputs, a last block of 32 when the outputs are <= 32 mod 64
    int wino6_min_pixels = 16384;  // ... below that the F(4x4) kernel (items of 12 x 48 pixels quantise small maps badly)
    int wino1d = 0;                // F(4x4)-class layers (stride 1, more than 32 outputs, maps of at least wino4_min_pixels pixels): 1 = one-dimensional
                                   // Winograd F(4,3) on the bf16 matrix pipe with exactly split fp32 operands, loader / consumer persistent blocks
                                   // (b2f_w1b.hip); 0 = the fp32-MFMA F(4x4) kernel of rounds 1-4 (b2f_wino4.hip)
    int s2_tile_groups = 1;        // ... launches that cannot fill the chip: one output tile per block (conv3x3_s2b<1, 1>, ConvLaunch::nsplit); same bits
    int s2_loader = 1;             // stride-2 layers on the bf16 pipe (bf16_conv >= 1): 1 = those of at least 64 input channels on the loader / consumer kernel
                                   // that computes all outputs of a tile (b2f_s2b.hip), 2 = all of them, 0 = conv3x3_bf6 (b2f_convb.hip)
    int wino4_persistent = 1;      // F(4x4) kernel: 1 = persistent blocks (one per CU, K pipeline continues across tiles), 0 = one tile per block, > 1 = that many persistent blocks (tests)
    int s2_tiles_per_block = 0;    // direct stride-2 kernel: tiles chained per block (0 = launcher's rule; bit-identical either way)
    long long host_subbatch_pixels = 16ll << 20;
    int host_threads = 0;          // 0 = auto
    int host_u8 = 1, host_ramp = 1;
    int debug_fail_next = 0;       // tests: the next b2f_compute_flow* call on this context fails (cross-thread error hand-over of b2f_multi_*)
    std::map<b2f::GraphKey, hipGraphExec_t> graphs;
    // output table of the generic executor (non-shipped graph shapes), kept between calls of one (B, H, W)
    std::vector<float *> gen_out;
    int genB = 0, genH = 0, genW = 0;
    // profiling
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<long long> prof_n;
    std::vector<b2f::ProfEvent> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    // host-buffer pipeline (b2f_compute_flow_batch)
    hipStream_t s_in = nullptr, s_out = nullptr;
    b2f::HostSlot slot[2];
    std::unique_ptr<b2f::CopyPool> pool_in, pool_out;
};

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return b2f::api_fail(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)
// function-try-block tail of every int-returning entry point: nothing is thrown across the C ABI
#define B2F_CATCH(fn_)                                                                        \
    catch (const std::exception &e_) { return b2f::api_fail(std::string(fn_ ": ") + e_.what()); } \
    catch (...) { return b2f::api_fail(fn_ ": unknown exception"); }
#define CHK(expr)                         \
    do {                                  \
        int rc_ = (expr);                 \
        if (rc_ != 0) return rc_;         \
    } while (0)

namespace b2f {
// b2f_api.hip
int check_shape(int B, int H, int W);
void drop_graphs(b2f_ctx *c);
void drop_gen_out(b2f_ctx *c);
// model:forward on device pointers, optionally replayed from a hipGraph (see b2f_api.hip)
int forward_device(b2f_ctx *c, const void *dev_in, int in_kind, int B, int H, int W, float *dev_flow, float *dev_occ,
                   float *dev_est3, hipStream_t s, bool graph);
// pieces of b2f_api.hip the generic graph executor (b2f_graph.hip) builds on
ConvSeg cp8_seg(const float *ptr, int C, size_t hw);
int find_conv_id(const b2f_ctx *c, int kind, int level, int idx);
int run_conv_layer(b2f_ctx *c, hipStream_t s, bool cap, int conv_id, const ConvSeg *segs, int nimg, int H, int W, int stride, int leaky,
                   float *out);
int ensure_arena(b2f_ctx *c, size_t floats);
// b2f_graph.hip: model:forward for the non-shipped graph shapes; outs = the whole output table (planar device buffers)
int graph_forward(b2f_ctx *c, hipStream_t s, bool cap, const void *dev_in, int in_kind, int B, int H, int W, float *const *outs);
}  // namespace b2f
