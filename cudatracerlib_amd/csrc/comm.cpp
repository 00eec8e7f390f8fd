// comm.cpp — the one collective of a multi-GPU render inside the library (BASELINE north_star: "a single RCCL gather of the
// framebuffer over xGMI at the end of each pass"): every rank renders its image tiles into a zero-initialised full-size PixelData frame, so
// ncclReduce(sum) to the root IS the gather.  A C++ host (the reference's main.cpp:160-172 flow) uses 8 GPUs through these four calls; the
// unique id travels between the ranks by whatever the host has (bench.py: torch.distributed over gloo; an MPI host: MPI_Bcast).
// The reference has nothing here (single device).  RCCL is loaded with dlopen at the first call, so that the library itself — and the CPU-only
// test suite — does not depend on librccl.so being loadable.
#include "tracer.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>
#include <string>

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
    ncclComm_t comm = nullptr; int rank = 0, world = 1; hipStream_t stream = nullptr;
    ~Comm() { if (comm) (void)rccl().CommDestroy(comm); if (stream) (void)hipStreamDestroy(stream); }
};

void comm_unique_id(unsigned char out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ctl_comm_get_unique_id hands out NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id; check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, 128);
}
Comm* comm_create(const unsigned char id_bytes[128], int rank, int world) {
    require_device();
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ctl_comm_create: bad rank / world");
    std::unique_ptr<Comm> c(new Comm());
    c->rank = rank; c->world = world;
    ncclUniqueId id; std::memcpy(&id, id_bytes, 128);
    check(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");   // one rank per process, on the process's current device
    CTL_HIP(hipStreamCreate(&c->stream));
    return c.release();
}
void comm_destroy(Comm* c) { delete c; }
// sum of all ranks' PixelData frames into the root's image (in place); the other ranks' images are left as they were.  Returns when the reduce is complete.
void comm_reduce_image(Comm* c, Image* img, int root) {
    if (root < 0 || root >= c->world) throw std::runtime_error("ctl_image_reduce: bad root");
    CTL_HIP(hipDeviceSynchronize());   // the frame is complete (the tracer renders on its own stream)
    const size_t n = (size_t)img->getWidth() * img->getHeight() * (sizeof(ctl_pixel_data) / sizeof(float));
    static_assert(sizeof(ctl_pixel_data) == 7 * sizeof(float), "PixelData is seven floats");
    check(rccl().Reduce(img->device(), img->device(), n, ncclFloat, ncclSum, root, c->comm, c->stream), "ncclReduce");
    CTL_HIP(hipStreamSynchronize(c->stream));
}

}  // namespace ctl
