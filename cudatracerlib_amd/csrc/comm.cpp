// comm.cpp — the one collective of a multi-GPU render inside the library (BASELINE north_star: "a single RCCL gather of the
// framebuffer over xGMI at the end of each pass"): every rank renders its image tiles into a zero-initialised full-size PixelData frame, so
// ncclReduce(sum) to the root IS the gather.  A C++ host (the reference's main.cpp:160-172 flow) uses 8 GPUs through these four calls; the
// unique id travels between the ranks by whatever the host has (bench.py: torch.distributed over gloo; an MPI host: MPI_Bcast).
// The reference has nothing here (single device).  RCCL is loaded with dlopen at the first call, so that the library itself — and the CPU-only
// test suite — does not depend on librccl.so being loadable.
#include "tracer.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdlib>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

namespace ctl {

namespace {
struct rccl_api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
const rccl_api& rccl() {
    static rccl_api api; static std::once_flag once; static std::string err;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { err = std::string("RCCL is not loadable: ") + (dlerror() ? dlerror() : "librccl.so not found"); return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.Reduce = (decltype(api.Reduce))dlsym(h, "ncclReduce");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Reduce) err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclReduce";
    });
    if (!err.empty()) throw std::runtime_error(err);
    return api;
}
void check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return;
    const rccl_api& a = rccl();
    throw std::runtime_error(std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error " + std::to_string((int)r)));
}
}  // namespace

struct Comm {
    ncclComm_t comm = nullptr; int rank = 0, world = 1; hipStream_t stream = nullptr; hipEvent_t done = nullptr;
    ~Comm() { if (comm) (void)rccl().CommDestroy(comm); if (done) (void)hipEventDestroy(done); if (stream) (void)hipStreamDestroy(stream); }
};
// how long a collective call may take before it is given up (ms; ctl_comm_create_timeout's argument, else $CTL_COMM_TIMEOUT_MS, else 120 s).  A rank that never arrives makes
// ncclCommInitRank / ncclReduce wait for ever: the caller gets an error it can act on (bench.py falls back to torch.distributed on every rank) instead of a hung job.
static int default_timeout_ms() { const char* e = std::getenv("CTL_COMM_TIMEOUT_MS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 120000; }

void comm_unique_id(unsigned char out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ctl_comm_get_unique_id hands out NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id; check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, 128);
}
Comm* comm_create(const unsigned char id_bytes[128], int rank, int world, int timeout_ms) {
    require_device();
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ctl_comm_create: bad rank / world");
    if (timeout_ms <= 0) timeout_ms = default_timeout_ms();
    const rccl_api& api = rccl();
    int dev = 0; CTL_HIP(hipGetDevice(&dev));
    ncclUniqueId id; std::memcpy(&id, id_bytes, 128);
    // ncclCommInitRank is collective and has no deadline of its own: it runs on a helper thread (on this thread's device) and is waited for
    struct pending { std::promise<std::pair<ncclResult_t, ncclComm_t>> p; };
    auto st = std::make_shared<pending>();
    auto fut = st->p.get_future();
    std::thread([st, &api, id, rank, world, dev]() {
        ncclComm_t c = nullptr; ncclResult_t r = ncclSystemError;
        if (hipSetDevice(dev) == hipSuccess) r = api.CommInitRank(&c, world, id, rank);   // one rank per process, on the process's current device
        st->p.set_value({ r, c });
    }).detach();
    if (fut.wait_for(std::chrono::milliseconds(timeout_ms)) != std::future_status::ready)
        throw std::runtime_error("ctl_comm_create: ncclCommInitRank of rank " + std::to_string(rank) + " / " + std::to_string(world) + " did not return within " + std::to_string(timeout_ms) + " ms (a rank that never arrived?)");
    const auto res = fut.get();
    check(res.first, "ncclCommInitRank");
    std::unique_ptr<Comm> c(new Comm());
    c->rank = rank; c->world = world; c->comm = res.second;
    CTL_HIP(hipStreamCreate(&c->stream));
    CTL_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    return c.release();
}
void comm_destroy(Comm* c) { delete c; }
// sum over the ranks of `src` (every rank's own PixelData frame) into `dst` on the root; dst == src is the in-place form.  dst is only read on the root (may be null elsewhere).
// Returns when the reduce is complete, or throws when it did not complete within the time-out.
void comm_reduce_image(Comm* c, Image* src, Image* dst, int root) {
    if (root < 0 || root >= c->world) throw std::runtime_error("ctl_image_reduce: bad root");
    if (c->rank == root && !dst) throw std::runtime_error("ctl_image_reduce_to: the root needs a destination image");
    if (dst && (dst->getWidth() != src->getWidth() || dst->getHeight() != src->getHeight())) throw std::runtime_error("ctl_image_reduce_to: source and destination sizes differ");
    if (dst == src && src->holds_reduced_frame())
        throw std::runtime_error("ctl_image_reduce: this image already holds a reduced frame — a second in-place reduce would add the other ranks' tiles onto sums that contain them; "
                                 "use ctl_image_reduce_to for a per-pass (progressive) gather, or clear the image first");
    CTL_HIP(hipDeviceSynchronize());   // the frame is complete (the tracer renders on its own stream)
    const size_t n = (size_t)src->getWidth() * src->getHeight() * (sizeof(ctl_pixel_data) / sizeof(float));
    static_assert(sizeof(ctl_pixel_data) == 7 * sizeof(float), "PixelData is seven floats");
    check(rccl().Reduce(src->device(), dst ? dst->device() : src->device(), n, ncclFloat, ncclSum, root, c->comm, c->stream), "ncclReduce");
    CTL_HIP(hipEventRecord(c->done, c->stream));
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(default_timeout_ms());
    for (;;) {
        const hipError_t q = hipEventQuery(c->done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) CTL_HIP(q);
        if (std::chrono::steady_clock::now() > deadline) throw std::runtime_error("ctl_image_reduce: ncclReduce did not complete within " + std::to_string(default_timeout_ms()) + " ms");
        if (std::chrono::steady_clock::now() + std::chrono::milliseconds(default_timeout_ms()) - deadline > std::chrono::milliseconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // spin for the first 5 ms (a reduce takes ~1), then poll
    }
    if (dst == src && c->rank == root) src->mark_reduced(true);
}

}  // namespace ctl
