// Multi-GPU entry points of the C ABI (include/b2f.h, "one node, several GPUs"): what the reference gets from
// nn.DataParallelTable (util.lua:27-48: one replica per GPU, each driven by its own host thread, :34-40; batch split
// along dimension 1, :32; parameters synchronised with NCCL, train.lua:494-496) for the inference path, usable from
// the LuaJIT / C host without Python or torch:
//   * one b2f_ctx per GPU inside one process, one worker thread per GPU;
//   * the weights are loaded / generated once (GPU 0) and broadcast into every other context's flat device buffer
//     with RCCL (ncclBroadcast inside one ncclGroup, over xGMI between the GPUs of a node; the library is dlopen'ed
//     so that libb2f.so has no link-time dependency on it) or, if RCCL cannot be loaded, with hipMemcpyPeer;
//   * a batch of independent triplets is split statically and contiguously over the GPUs (sizes differ by at most
//     one, earlier GPUs take the remainder) and every worker writes straight into its slice of the caller's output
//     buffers.  No data-path collective: computeFlow keeps no cross-sample state.
#include "b2f_ctx.h"

#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace b2f;

struct b2f_multi {
    std::vector<b2f_ctx *> ctx;
    std::vector<int> devices;
    int transport = 0;   // 0 = single GPU (nothing to copy), 1 = RCCL broadcast, 2 = hipMemcpyPeer
};

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        // RTLD_NOLOAD first: a host that already carries an RCCL (e.g. PyTorch's bundled copy) keeps using that one
        for (int pass = 0; pass < 2 && !h; ++pass)
            for (const char *n : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) break;
            }
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
    }
};

// Broadcast context 0's flat weight buffer into the others, in place on their device buffers, and rebuild their packed
// copies.  *transport: in = preferred (1 RCCL, 2 peer copy), out = used.
int broadcast_weights(b2f_multi *m, int *transport)
{
    const int n = (int)m->ctx.size();
    void *src = nullptr;
    long long count = 0;
    CHK(b2f_weights_device(m->ctx[0], &src, &count));
    std::vector<void *> dst((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {
        long long ci = 0;
        CHK(b2f_weights_device(m->ctx[i], &dst[(size_t)i], &ci));
        if (ci != count) return api_fail("b2f_init_multi: replicas disagree on the parameter count");
    }
    Rccl r;
    bool done = false;
    std::string why;
    if (*transport == 1 && r.load()) {
        // Communicators and the open group are released on every path (an early return between GroupStart and GroupEnd /
        // CommDestroy would leak all n communicators and leave RCCL's group open for the next rebroadcast): errors are
        // collected, the group is always closed, the communicators always destroyed, and ANY failure falls back to peer copies.
        std::vector<ncclComm_t> comms((size_t)n, nullptr);
        ncclResult_t rc = r.CommInitAll(comms.data(), n, m->devices.data());
        hipError_t he = hipSuccess;
        if (rc == ncclSuccess) {
            rc = r.GroupStart();
            const bool group_open = rc == ncclSuccess;
            for (int i = 0; i < n && rc == ncclSuccess && he == hipSuccess; ++i) {
                he = hipSetDevice(m->devices[(size_t)i]);
                if (he == hipSuccess)
                    rc = r.Broadcast(src, dst[(size_t)i], (size_t)count, ncclFloat, 0, comms[(size_t)i], m->ctx[(size_t)i]->stream);
            }
            if (group_open) {
                const ncclResult_t re = r.GroupEnd();
                if (rc == ncclSuccess) rc = re;
            }
            for (int i = 0; i < n; ++i) {          // drain every stream before the communicators go away, whatever happened
                hipError_t e1 = hipSetDevice(m->devices[(size_t)i]);
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(m->ctx[(size_t)i]->stream);
                if (he == hipSuccess) he = e1;
            }
        }
        for (ncclComm_t c : comms)
            if (c) (void)r.CommDestroy(c);
        if (rc == ncclSuccess && he == hipSuccess) done = true;
        else why = rc != ncclSuccess ? std::string("RCCL: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?") : std::string("HIP: ") + hipGetErrorString(he);
        // a HIP error here (stream sync / set device) is a faulted device, not a transport problem: no peer copies on top of it
        if (he != hipSuccess) return api_fail("b2f_init_multi: weight broadcast failed (" + why + ")");
        if (!done) fprintf(stderr, "libb2f: RCCL weight broadcast failed (%s); falling back to hipMemcpyPeer\n", why.c_str());
    } else if (*transport == 1) {
        fprintf(stderr, "libb2f: librccl.so not loadable; weight broadcast by hipMemcpyPeer\n");
    }
    if (!done) {
        *transport = 2;
        for (int i = 1; i < n; ++i) {
            if (m->devices[(size_t)i] == m->devices[0]) {
                // two replicas on one GPU (test switch B2F_MULTI_ALLOW_DUPLICATE): a peer copy onto the same device
                HIPCHK(hipSetDevice(m->devices[0]));
                HIPCHK(hipMemcpy(dst[(size_t)i], src, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice));
            } else {
                HIPCHK(hipMemcpyPeer(dst[(size_t)i], m->devices[(size_t)i], src, m->devices[0], (size_t)count * sizeof(float)));
            }
        }
    }
    for (int i = 1; i < n; ++i) CHK(b2f_commit_weights(m->ctx[(size_t)i]));
    return 0;
}

// One worker thread per GPU with a non-empty shard; `call(i, lo, hi)` runs on it.
template <class F>
int run_sharded(b2f_multi *m, int n, F call)
{
    const int g = (int)m->ctx.size();
    std::vector<int> rc((size_t)g, 0);
    std::vector<std::string> err((size_t)g);
    std::vector<std::thread> th;
    for (int i = 0; i < g; ++i) {
        int lo = 0, hi = 0;
        (void)b2f_shard_range(n, i, g, &lo, &hi);
        if (hi <= lo) continue;
        th.emplace_back([&, i, lo, hi] {
            rc[(size_t)i] = call(i, lo, hi);
            if (rc[(size_t)i]) err[(size_t)i] = api_error();   // the message is thread-local: hand it to the caller's thread
        });
    }
    for (std::thread &t : th) t.join();
    for (int i = 0; i < g; ++i)
        if (rc[(size_t)i]) return api_fail("GPU " + std::to_string(m->devices[(size_t)i]) + ": " + err[(size_t)i]);
    return 0;
}

}  // namespace

extern "C" {

int b2f_shard_range(int n, int rank, int world, int *lo, int *hi) try
{
    if (n < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return api_fail("b2f_shard_range: bad arguments");
    const int q = n / world, r = n % world;
    *lo = rank * q + std::min(rank, r);
    *hi = *lo + q + (rank < r ? 1 : 0);
    return 0;
}
B2F_CATCH("b2f_shard_range")

int b2f_init_multi(const char *name_or_path, int n_gpus, const int *devices, b2f_multi **out) try
{
    if (!out) return api_fail("b2f_init_multi: null out");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return api_fail("b2f_init_multi: no HIP device available (this library has no CPU fallback)");
    if (n_gpus == 0) n_gpus = ndev;   // all visible GPUs
    // Test switch (tests/test_gpu_parity.py: the N > 1 code path on a one-GPU box): B2F_MULTI_ALLOW_DUPLICATE=1 together with
    // B2F_MULTI_TRANSPORT=peer lets a device be listed more than once -- several replicas, worker threads and shards on one GPU
    const char *tr_env = getenv("B2F_MULTI_TRANSPORT"), *dup_env = getenv("B2F_MULTI_ALLOW_DUPLICATE");
    const bool allow_dup = dup_env && atoi(dup_env) == 1 && tr_env && !strcmp(tr_env, "peer");
    if (n_gpus < 0 || (n_gpus > ndev && !allow_dup) || n_gpus > 64) return api_fail("b2f_init_multi: n_gpus exceeds the visible devices");
    std::unique_ptr<b2f_multi> m(new b2f_multi());
    struct Guard {
        b2f_multi *m;
        ~Guard() { if (m) for (b2f_ctx *c : m->ctx) b2f_destroy(c); }
    } guard{m.get()};
    for (int i = 0; i < n_gpus; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= ndev) return api_fail("b2f_init_multi: bad device ordinal");
        for (int e : m->devices)
            if (e == d && !allow_dup) return api_fail("b2f_init_multi: a device is listed twice");
        m->devices.push_back(d);
    }
    // replica 0 reads the file / draws the weights; the others start from a different deterministic set of the same
    // architecture, so that only the broadcast can make them agree (b2f_multi_weights_checksum shows it)
    b2f_ctx *c0 = nullptr;
    CHK(b2f_init(name_or_path, m->devices[0], &c0));
    m->ctx.push_back(c0);
    for (int i = 1; i < n_gpus; ++i) {
        b2f_ctx *ci = nullptr;
        const std::string filler = std::string("random:") + (c0->past_flow ? "soft:" : "hard:") + std::to_string(7000 + i);
        CHK(b2f_init(filler.c_str(), m->devices[(size_t)i], &ci));
        m->ctx.push_back(ci);
    }
    const char *e = tr_env;   // "peer": skip RCCL; "selftest": run the broadcast even on one GPU
    if (n_gpus > 1 || (e && !strcmp(e, "selftest"))) {
        int transport = (e && !strcmp(e, "peer")) ? 2 : 1;
        CHK(broadcast_weights(m.get(), &transport));
        m->transport = transport;
    }
    guard.m = nullptr;
    *out = m.release();
    return 0;
}
B2F_CATCH("b2f_init_multi")

void b2f_destroy_multi(b2f_multi *m)
{
    if (!m) return;
    for (b2f_ctx *c : m->ctx) b2f_destroy(c);
    delete m;
}

int b2f_multi_info(const b2f_multi *m, int *n_gpus, int *devices, int cap, int *transport) try
{
    if (!m) return api_fail("b2f_multi_info: null context");
    if (n_gpus) *n_gpus = (int)m->ctx.size();
    if (devices)
        for (int i = 0; i < cap && i < (int)m->devices.size(); ++i) devices[i] = m->devices[(size_t)i];
    if (transport) *transport = m->transport;
    return 0;
}
B2F_CATCH("b2f_multi_info")

b2f_ctx *b2f_multi_context(b2f_multi *m, int i)
{
    return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr;
}

int b2f_multi_rebroadcast(b2f_multi *m) try
{
    if (!m) return api_fail("b2f_multi_rebroadcast: null context");
    if (m->ctx.size() < 2) return 0;
    int transport = m->transport == 2 ? 2 : 1;
    CHK(broadcast_weights(m, &transport));
    m->transport = transport;
    return 0;
}
B2F_CATCH("b2f_multi_rebroadcast")

// FNV-1a over the bits of every replica's flat weight buffer as it sits on its GPU
int b2f_multi_weights_checksum(b2f_multi *m, unsigned long long *sums, int cap) try
{
    if (!m || !sums) return api_fail("b2f_multi_weights_checksum: null argument");
    for (int i = 0; i < (int)m->ctx.size() && i < cap; ++i) {
        std::vector<float> w((size_t)m->ctx[(size_t)i]->nparams);
        CHK(b2f_get_weights(m->ctx[(size_t)i], w.data(), (long long)w.size()));
        unsigned long long h = 1469598103934665603ull;
        const unsigned char *p = (const unsigned char *)w.data();
        for (size_t k = 0; k < w.size() * sizeof(float); ++k) { h ^= p[k]; h *= 1099511628211ull; }
        sums[i] = h;
    }
    return 0;
}
B2F_CATCH("b2f_multi_weights_checksum")

int b2f_multi_compute_flow_batch(b2f_multi *m, int n, const float *im1, const float *im2, const float *im3, int H0, int W0,
                                 double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ) try
{
    if (!m || !im1 || !im2 || !im3 || !flow || !fwd_occ || !bwd_occ) return api_fail("b2f_multi_compute_flow_batch: null argument");
    if (n <= 0 || H0 <= 0 || W0 <= 0) return api_fail("b2f_multi_compute_flow_batch: bad shape");
    const size_t hw = (size_t)H0 * W0;
    return run_sharded(m, n, [&](int i, int lo, int hi) {
        return b2f_compute_flow_batch(m->ctx[(size_t)i], hi - lo, im1 + (size_t)lo * 3 * hw, im2 + (size_t)lo * 3 * hw, im3 + (size_t)lo * 3 * hw,
                                      H0, W0, flow + (size_t)lo * 2 * hw, fwd_occ + (size_t)lo * hw, bwd_occ + (size_t)lo * hw);
    });
}
B2F_CATCH("b2f_multi_compute_flow_batch")

int b2f_multi_compute_flow_batch_u8(b2f_multi *m, int n, const unsigned char *im1, const unsigned char *im2, const unsigned char *im3,
                                    int H0, int W0, double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ) try
{
    if (!m || !im1 || !im2 || !im3 || !flow || !fwd_occ || !bwd_occ) return api_fail("b2f_multi_compute_flow_batch_u8: null argument");
    if (n <= 0 || H0 <= 0 || W0 <= 0) return api_fail("b2f_multi_compute_flow_batch_u8: bad shape");
    const size_t hw = (size_t)H0 * W0;
    return run_sharded(m, n, [&](int i, int lo, int hi) {
        return b2f_compute_flow_batch_u8(m->ctx[(size_t)i], hi - lo, im1 + (size_t)lo * 3 * hw, im2 + (size_t)lo * 3 * hw, im3 + (size_t)lo * 3 * hw,
                                         H0, W0, flow + (size_t)lo * 2 * hw, fwd_occ + (size_t)lo * hw, bwd_occ + (size_t)lo * hw);
    });
}
B2F_CATCH("b2f_multi_compute_flow_batch_u8")

}  // extern "C"
