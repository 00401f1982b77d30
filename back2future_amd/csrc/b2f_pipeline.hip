// Host-buffer boundary of libb2f.so: b2f_compute_flow / b2f_compute_flow_batch[_u8] (back2future.lua:47-95) as a
// double-buffered upload / kernels / download pipeline around model:forward.  Device kernels of the pre/post
// processing: b2f_boundary.hip.
#include "b2f_ctx.h"

#include <cstdlib>

using namespace b2f;

static int fail(const std::string &m) { return api_fail(m); }

// ---- host-buffer entry point: a double-buffered upload / compute / download pipeline ----------------
namespace {

// 1 when [p, p + bytes) is page-locked host memory known to the HIP runtime (hipHostMalloc / hipHostRegister, e.g. a
// torch pin_memory() tensor): such buffers are DMA'd directly; 0 for ordinary (pageable) host memory, which goes
// through the pinned slot; -1 for device / managed memory, which the host entry points refuse
int mem_kind(const void *p, size_t bytes)
{
    int kind = 1;
    for (const char *q : {(const char *)p, (const char *)p + bytes - 1}) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) {
            (void)hipGetLastError();
            kind = 0;
            continue;
        }
        if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeArray) return -1;
        if (a.type != hipMemoryTypeHost) kind = 0;
    }
    return kind;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// carve the slot's device and pinned blobs for sub-batches of up to SB triplets; grows (never shrinks) the blobs
int ensure_slot(b2f_ctx *c, HostSlot &hs, int SB, size_t hw0, size_t hw, int H0, int fw, bool same, int C3, bool stage_in,
                bool stage_masks, bool use_u8)
{
    const size_t n_up = align256((size_t)SB * 9 * hw0 * 4), n_u8 = use_u8 ? align256((size_t)SB * 9 * hw0) : 0,
                 n_in = same ? 0 : align256((size_t)SB * 9 * hw * 4), n_tmp = same ? 0 : align256((size_t)SB * 9 * H0 * fw * 4),
                 n_flow = align256((size_t)SB * 2 * hw * 4), n_est3 = align256((size_t)SB * C3 * hw * 4),
                 n_f32 = align256((size_t)SB * 2 * hw0 * 4), n_occ = align256((size_t)SB * hw0);
    const size_t need_dev = n_up + n_u8 + n_in + n_tmp + n_flow + n_est3 + (same ? 0 : n_f32) + 2 * n_occ;
    if (need_dev > hs.dev_bytes) {
        if (hs.dev) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipFree(hs.dev));
            hs.dev = nullptr; hs.dev_bytes = 0;
            drop_graphs(c);   // graphs are keyed on slot pointers
        }
        HIPCHK(hipMalloc(&hs.dev, need_dev));
        hs.dev_bytes = need_dev;
    }
    char *d = hs.dev;
    hs.d_up = (float *)d; d += n_up;
    hs.d_u8 = (unsigned char *)d; d += n_u8;
    hs.d_in = same ? hs.d_up : (float *)d; d += n_in;
    hs.d_tmp = (float *)d; d += n_tmp;
    hs.d_flow = (float *)d; d += n_flow;
    hs.d_est3 = (float *)d; d += n_est3;
    hs.d_flow32 = same ? hs.d_flow : (float *)d; d += same ? 0 : n_f32;
    hs.d_fo = (unsigned char *)d; d += n_occ;
    hs.d_bo = (unsigned char *)d;
    const size_t need_pin = (stage_in ? n_up : 0) + n_u8 + n_f32 + (stage_masks ? 2 * n_occ : 0);
    if (need_pin > hs.pin_bytes) {
        if (hs.pin) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipHostFree(hs.pin));
            hs.pin = nullptr; hs.pin_bytes = 0;
        }
        HIPCHK(hipHostMalloc(&hs.pin, need_pin, hipHostMallocDefault));
        hs.pin_bytes = need_pin;
    }
    char *h = hs.pin;
    hs.h_in = (float *)h; h += stage_in ? n_up : 0;
    hs.h_u8 = (unsigned char *)h; h += n_u8;
    hs.h_flow32 = (float *)h; h += n_f32;
    hs.h_fo = (unsigned char *)h; h += stage_masks ? n_occ : 0;
    hs.h_bo = (unsigned char *)h;
    for (hipEvent_t *e : {&hs.ev_in, &hs.ev_comp, &hs.ev_out})
        if (!*e) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return 0;
}

}  // namespace

// The n triplets are cut into sub-batches (up to B2F_HOST_SUBBATCH_PIXELS input pixels each, default 16 Mpx = eight
// full-HD triplets) that flow through two buffer sets on three streams: uploads on s_in, ColorNormalize /
// image.scale / the network / the nearest rescale + thresholds on the context's stream, downloads on s_out.
// A buffer set's input half is reused as soon as the kernels that read it are done and its output half as soon
// as its download has been handed over, so in steady state all three streams are busy.
// The link is the bound of this entry point (one MI355X box: 56 GB/s in either direction, but only 55 GB/s for
// both together), so both directions carry as few bytes as exactness allows: inputs that are k / 255 go up as
// bytes (pack_u8_piece), the flow comes down as the network's fp32 values and becomes `double * sc` on the host
// (back2future.lua:80-84), exactly the reference's arithmetic.  Host threads (option host_threads, default 16, two
// thirds on the input side; the output side is driven by a second control thread) do the packing / staging and
// the f32 -> f64 conversion; page-locked caller buffers are DMA'd in place where no conversion is involved.
namespace {
int compute_flow_pipeline(b2f_ctx *c, int n, const void *im1, const void *im2, const void *im3, bool bytes_in, int H0,
                          int W0, double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ)
{
    if (!c || !im1 || !im2 || !im3 || !flow || !fwd_occ || !bwd_occ) return fail("b2f_compute_flow: null argument");
    if (n <= 0 || H0 <= 0 || W0 <= 0) return fail("b2f_compute_flow: bad shape");
    if (c->debug_fail_next) {   // tests (option debug_fail_next): one forced failure, e.g. on one replica of a b2f_multi
        c->debug_fail_next = 0;
        return fail("b2f_compute_flow: forced failure (option debug_fail_next)");
    }
    const int fw = W0 - W0 % 64, fh = H0 - H0 % 64;   // back2future.lua:54-67
    if (fw <= 0 || fh <= 0) return fail("b2f_compute_flow: image smaller than 64 pixels");
    CHK(check_shape(1, fh, fw));
    // the kernel choice of every sub-batch follows the caller's n (a single-triplet request takes the per-launch rule, a batch the map-size
    // rule -- for ALL its sub-batches, also those of one triplet the ramp and the tail produce)
    struct ReqBatch { b2f_ctx *c; ReqBatch(b2f_ctx *cc, int nn) : c(cc) { c->req_batch = nn; } ~ReqBatch() { c->req_batch = 0; } } req_guard(c, n);
    HIPCHK(hipSetDevice(c->device));
    const size_t hw0 = (size_t)H0 * W0, hw = (size_t)fh * fw;
    const bool same = (fw == W0 && fh == H0);
    const int C3 = c->past_flow ? 2 : 3;
    const double sc_h = (double)H0 / (double)fh, sc_w = (double)W0 / (double)fw;   // :78-79
    const long long sub_px = c->host_subbatch_pixels;
    const int nthreads = std::max(2, c->host_threads > 0 ? c->host_threads : (int)std::min(16u, std::thread::hardware_concurrency()));
    const bool use_u8 = bytes_in || c->host_u8 != 0;
    const size_t esz = bytes_in ? 1 : 4;   // bytes per input sample in the caller's buffers
    const int SB = (int)std::min<long long>(n, std::max<long long>(1, sub_px / (long long)hw0));
    // sub-batch sizes ramp up from ~2 Mpx (one full-HD triplet) by doubling to SB: the kernels start after a small
    // upload instead of SB triplets' (option host_ramp = 0: uniform sizes)
    const bool ramp = c->host_ramp != 0;
    const int sz0 = ramp ? (int)std::min<long long>(SB, std::max<long long>(1, (2ll << 20) / (long long)hw0)) : SB;
    std::vector<std::pair<size_t, int>> subs;   // (first triplet, count)
    for (int b0 = 0, sz = sz0; b0 < n; sz = std::min(2 * sz, SB)) {
        const int nb = std::min(sz, n - b0);
        subs.push_back({(size_t)b0, nb});
        b0 += nb;
    }
    const int nsub = (int)subs.size();

    if (!c->s_in) HIPCHK(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    if (!c->s_out) HIPCHK(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    const int k_in[3] = {mem_kind(im1, (size_t)n * 3 * hw0 * esz), mem_kind(im2, (size_t)n * 3 * hw0 * esz),
                         mem_kind(im3, (size_t)n * 3 * hw0 * esz)};
    const int k_out[3] = {mem_kind(flow, (size_t)n * 2 * hw0 * 8), mem_kind(fwd_occ, (size_t)n * hw0), mem_kind(bwd_occ, (size_t)n * hw0)};
    for (int i = 0; i < 3; ++i)
        if (k_in[i] < 0 || k_out[i] < 0)
            return fail("b2f_compute_flow: device memory passed to a host-buffer entry point (use b2f_forward_device)");
    const bool pinned_in = k_in[0] == 1 && k_in[1] == 1 && k_in[2] == 1;
    const bool stage_in = !pinned_in && !bytes_in;   // float staging buffer (byte inputs stage through h_u8)
    const bool stage_masks = !(k_out[1] == 1 && k_out[2] == 1);
    for (int k = 0; k < std::min(nsub, 2); ++k)
        CHK(ensure_slot(c, c->slot[k], SB, hw0, hw, H0, fw, same, C3, stage_in, stage_masks, use_u8));
    // the calling thread and the drain thread each count as one worker of their pool
    const int w_out = std::max(0, nthreads / 3 - 1), w_in = std::max(0, nthreads - nthreads / 3 - 1);
    if (!c->pool_in || c->pool_in->workers() != w_in) c->pool_in.reset(new CopyPool(w_in));
    if (!c->pool_out || c->pool_out->workers() != w_out) c->pool_out.reset(new CopyPool(w_out));

    const char *ims[3] = {(const char *)im1, (const char *)im2, (const char *)im3};
    // ---- output side: a second control thread hands finished downloads to the caller ----
    std::mutex mu;
    std::condition_variable cv;
    int submitted = 0, drained = 0;   // sub-batches whose downloads are enqueued / handed over (guarded by mu)
    bool abort = false;
    std::string drain_err;
    auto drain_loop = [&]() {
        (void)hipSetDevice(c->device);
        for (int k = 0; k < nsub; ++k) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return submitted > k || abort; });
                if (abort) return;
            }
            HostSlot &hs = c->slot[k & 1];
            const hipError_t e = hipEventSynchronize(hs.ev_out);
            if (e == hipSuccess) {
                const size_t b0 = subs[k].first, nb = (size_t)subs[k].second;
                std::vector<CopyJob> jobs;
                // flow_est[1] * sc_w, flow_est[2] * sc_h on the :double() copy of est[1] (:80-84)
                for (size_t t = 0; t < nb; ++t)
                    for (int ch = 0; ch < 2; ++ch)
                        jobs.push_back({flow + ((b0 + t) * 2 + ch) * hw0, hs.h_flow32 + (t * 2 + ch) * hw0, hw0 * 4, JOB_F32_TO_F64,
                                        ch == 0 ? sc_w : sc_h, nullptr});
                if (stage_masks) {
                    jobs.push_back({fwd_occ + b0 * hw0, hs.h_fo, nb * hw0});
                    jobs.push_back({bwd_occ + b0 * hw0, hs.h_bo, nb * hw0});
                }
                c->pool_out->run(jobs);
            }
            std::lock_guard<std::mutex> l(mu);
            if (e != hipSuccess) {
                drain_err = std::string("download failed: ") + hipGetErrorString(e);
                abort = true;
            }
            drained = k + 1;
            cv.notify_all();
            if (abort) return;
        }
    };
    std::thread drainer(drain_loop);
    struct Joiner {   // an exception on the way out (std::bad_alloc ...) must not leave the thread running
        std::thread &t; std::mutex &mu; std::condition_variable &cv; bool &abort;
        ~Joiner()
        {
            if (!t.joinable()) return;
            { std::lock_guard<std::mutex> l(mu); abort = true; }
            cv.notify_all();
            t.join();
        }
    } joiner{drainer, mu, cv, abort};

    bool try_u8 = use_u8;             // off for the rest of the call after the first triplet that is not 8-bit data
    std::atomic<int> inexact{0};
    auto submit = [&](int k) -> int {
        HostSlot &hs = c->slot[k & 1];
        const size_t b0 = subs[k].first;
        const int nb = subs[k].second;
        // ---- upload: torch.cat({im1, im2, im3}, 1) (back2future.lua:48) = [triplet][frame][3][H0][W0] on the device.
        // The set's staging buffers are free once upload k - 2 has left them, its device buffers once the kernels
        // of k - 2 are done (both events still hold the records of k - 2 here).
        if (k >= 2) HIPCHK(hipEventSynchronize(hs.ev_in));
        if (k >= 2) HIPCHK(hipStreamWaitEvent(c->s_in, hs.ev_comp, 0));
        std::vector<int> as_u8(nb, 0);
        for (int t = 0; t < nb; ++t) {
            float *dst = hs.d_up + (size_t)t * 9 * hw0;
            if (bytes_in) {   // the caller's samples are the bytes: value = k / 255
                unsigned char *du = hs.d_u8 + (size_t)t * 9 * hw0;
                if (pinned_in) {
                    for (int f = 0; f < 3; ++f)
                        HIPCHK(hipMemcpyAsync(du + (size_t)f * 3 * hw0, ims[f] + (b0 + t) * 3 * hw0, 3 * hw0, hipMemcpyHostToDevice, c->s_in));
                } else {
                    unsigned char *st = hs.h_u8 + (size_t)t * 9 * hw0;
                    for (int f = 0; f < 3; ++f) {   // frame by frame: the DMA of one frame runs under the staging copy of the next
                        c->pool_in->run({{st + (size_t)f * 3 * hw0, ims[f] + (b0 + t) * 3 * hw0, 3 * hw0}});
                        HIPCHK(hipMemcpyAsync(du + (size_t)f * 3 * hw0, st + (size_t)f * 3 * hw0, 3 * hw0, hipMemcpyHostToDevice, c->s_in));
                    }
                }
                as_u8[t] = 1;
                continue;
            }
            if (try_u8) {
                unsigned char *st = hs.h_u8 + (size_t)t * 9 * hw0;
                // frame by frame: the DMA of a packed frame runs under the packing of the next.  A frame that turns
                // out not to be 8-bit data sends the whole triplet down the float path (its earlier frames are
                // uploaded twice; d_up is what the kernels read for it).
                for (int f = 0; f < 3 && !inexact.load(); ++f) {
                    c->pool_in->run({{st + (size_t)f * 3 * hw0, ims[f] + (b0 + t) * 3 * hw0 * 4, 3 * hw0 * 4, JOB_PACK_U8, 1.0, &inexact}});
                    if (!inexact.load())
                        HIPCHK(hipMemcpyAsync(hs.d_u8 + ((size_t)t * 9 + (size_t)f * 3) * hw0, st + (size_t)f * 3 * hw0, 3 * hw0,
                                              hipMemcpyHostToDevice, c->s_in));
                }
                if (!inexact.load()) {
                    as_u8[t] = 1;
                    continue;
                }
                try_u8 = false;
            }
            if (stage_in) {
                float *st = hs.h_in + (size_t)t * 9 * hw0;
                for (int f = 0; f < 3; ++f) {
                    c->pool_in->run({{st + (size_t)f * 3 * hw0, ims[f] + (b0 + t) * 3 * hw0 * 4, 3 * hw0 * 4}});
                    HIPCHK(hipMemcpyAsync(dst + (size_t)f * 3 * hw0, st + (size_t)f * 3 * hw0, 3 * hw0 * 4, hipMemcpyHostToDevice, c->s_in));
                }
            } else {
                for (int f = 0; f < 3; ++f)
                    HIPCHK(hipMemcpyAsync(dst + (size_t)f * 3 * hw0, ims[f] + (b0 + t) * 3 * hw0 * 4, 3 * hw0 * 4, hipMemcpyHostToDevice, c->s_in));
            }
        }
        HIPCHK(hipEventRecord(hs.ev_in, c->s_in));
        // ---- kernels: after the upload, and after download k - 2 has read this set's output buffers
        HIPCHK(hipStreamWaitEvent(c->stream, hs.ev_in, 0));
        if (k >= 2) HIPCHK(hipStreamWaitEvent(c->stream, hs.ev_out, 0));
        for (int t = 0; t < nb; ++t)
            if (as_u8[t]) HIPCHK(launch_unpack_u8(hs.d_u8 + (size_t)t * 9 * hw0, 9 * hw0, hs.d_up + (size_t)t * 9 * hw0, c->stream));
        // ColorNormalize, then image.scale to the /64 size (:50-71); without a rescale the raw planes go to the
        // network as they are and the first conv kernel normalizes on the fly
        if (!same) HIPCHK(launch_image_scale(hs.d_up, 1, (long)nb * 9, H0, W0, hs.d_tmp, hs.d_in, fh, fw, c->stream));
        CHK(forward_device(c, hs.d_in, same ? B2F_IN_UNIT : B2F_IN_NORMALIZED, nb, fh, fw, hs.d_flow, nullptr, hs.d_est3, c->stream,
                           c->host_graph != 0));
        HIPCHK(launch_postprocess(hs.d_flow, hs.d_est3, C3, nb, fh, fw, H0, W0, same ? nullptr : hs.d_flow32, hs.d_fo, hs.d_bo, c->stream));
        HIPCHK(hipEventRecord(hs.ev_comp, c->stream));
        // ---- download: the set's pinned output buffers must have been handed over (k - 2 drained)
        if (k >= 2) {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return drained >= k - 1 || abort; });
            if (abort) return fail(drain_err);
        }
        HIPCHK(hipStreamWaitEvent(c->s_out, hs.ev_comp, 0));
        HIPCHK(hipMemcpyAsync(hs.h_flow32, hs.d_flow32, (size_t)nb * 2 * hw0 * 4, hipMemcpyDeviceToHost, c->s_out));
        HIPCHK(hipMemcpyAsync(stage_masks ? hs.h_fo : fwd_occ + b0 * hw0, hs.d_fo, (size_t)nb * hw0, hipMemcpyDeviceToHost, c->s_out));
        HIPCHK(hipMemcpyAsync(stage_masks ? hs.h_bo : bwd_occ + b0 * hw0, hs.d_bo, (size_t)nb * hw0, hipMemcpyDeviceToHost, c->s_out));
        HIPCHK(hipEventRecord(hs.ev_out, c->s_out));
        {
            std::lock_guard<std::mutex> l(mu);
            submitted = k + 1;
        }
        cv.notify_all();
        return 0;
    };
    int rc = 0;
    for (int k = 0; k < nsub && !rc; ++k) rc = submit(k);
    std::string msg = rc ? api_error() : std::string();
    if (rc) {
        std::lock_guard<std::mutex> l(mu);
        abort = true;
    }
    cv.notify_all();
    drainer.join();
    if (!rc && abort) { rc = 1; msg = drain_err; }
    // nothing of this call may still be in flight when the caller gets its buffers back
    for (hipStream_t st : {c->s_in, c->stream, c->s_out}) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess && !rc) { rc = 1; msg = std::string("b2f_compute_flow: ") + hipGetErrorString(e); }
    }
    if (rc) {
        (void)hipGetLastError();
        return fail(msg);
    }
    return 0;
}

}  // namespace

extern "C" {

int b2f_compute_flow_batch(b2f_ctx *c, int n, const float *im1, const float *im2, const float *im3, int H0,
                           int W0, double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ) try
{
    return compute_flow_pipeline(c, n, im1, im2, im3, false, H0, W0, flow, fwd_occ, bwd_occ);
}
B2F_CATCH("b2f_compute_flow_batch")

int b2f_compute_flow_batch_u8(b2f_ctx *c, int n, const unsigned char *im1, const unsigned char *im2,
                              const unsigned char *im3, int H0, int W0, double *flow, unsigned char *fwd_occ,
                              unsigned char *bwd_occ) try
{
    return compute_flow_pipeline(c, n, im1, im2, im3, true, H0, W0, flow, fwd_occ, bwd_occ);
}
B2F_CATCH("b2f_compute_flow_batch_u8")

int b2f_compute_flow(b2f_ctx *c, const float *im1, const float *im2, const float *im3, int H0, int W0,
                     double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ) try
{
    return b2f_compute_flow_batch(c, 1, im1, im2, im3, H0, W0, flow, fwd_occ, bwd_occ);
}
B2F_CATCH("b2f_compute_flow")

}  // extern "C"
