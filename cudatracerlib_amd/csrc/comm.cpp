// comm.cpp — the one collective of a multi-GPU render inside the library (BASELINE north_star: "a single RCCL gather of the
// framebuffer over xGMI at the end of each pass").  Rank r renders the 64x64 image tiles t with t % world == r into a full-size PixelData frame.
//   ctl_image_gather   — what north_star names: every rank packs ITS tiles with a one-pixel halo (ceil(tiles / world) x 65 x 65 x 28 B = 7.6 MB at 1080p / 8) and ONE
//                        ncclGather brings them to the root, which copies the tiles into the frame and adds the halos.  In place or out of place.
//   ctl_image_reduce   — the simpler equivalent kept as the agreed fallback: ncclReduce(sum) of the whole zero-padded frames (58 MB per rank at 1080p).
// A C++ host (the reference's main.cpp:160-172 flow) uses 8 GPUs through these calls; the unique id travels between the ranks by whatever the host
// has (bench.py: torch.distributed over gloo; an MPI host: MPI_Bcast).  A host without RCCL moves the same packed tiles itself (ctl_image_pack_tiles /
// ctl_image_unpack_tiles + MPI_Gather).  The reference has nothing here (single device).  RCCL is loaded with dlopen at the first call, so that the
// library itself — and the CPU-only test suite — does not depend on librccl.so being loadable.
#include "tracer.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <atomic>
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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;   // rccl.h:745 (an RCCL extension over send / recv)
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
        api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
        api.Reduce = (decltype(api.Reduce))dlsym(h, "ncclReduce");
        api.Gather = (decltype(api.Gather))dlsym(h, "ncclGather");      // optional: ctl_image_gather says so when it is absent and the host falls back to the reduce
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

// ---- the packed form of a rank's tiles: [slot k = tile k * world + rank][65 x 65 pixels, row-major][7 floats].  Rows / columns 0..63 of a slot are the tile itself;
// row 64 and column 64 are its HALO: the frame's pixels just right of and below the tile when they belong to ANOTHER rank (zero otherwise).  A sample's film position
// pixel + jitter rounds into the next pixel once in ~10^4 samples (1919 + 0.99999994f is 1920.0f, compaction.h add_sample_ordered — as in the reference, whose
// Image::AddSample floors the same position), and at a tile's right / bottom edge that next pixel is another rank's: the rank accumulated it in its own full-size frame,
// the halo carries it to the root, which ADDS it after all tiles are in place.  The clipped part of a border tile and a missing last slot are zero.
// One lane per float: the 455 consecutive floats of a slot row are 455 consecutive floats of the frame.
constexpr uint32_t kSlotEdge = 65u, kSlotPixels = kSlotEdge * kSlotEdge, kHaloPixels = 2u * kSlotEdge - 1u, kPixelFloats = sizeof(ctl_pixel_data) / sizeof(float);
static_assert(sizeof(ctl_pixel_data) == 7 * sizeof(float), "PixelData is seven floats");
__host__ __device__ inline uint32_t tile_count(uint32_t W, uint32_t H) { return ((W + 63u) / 64u) * ((H + 63u) / 64u); }
inline uint32_t slot_count(uint32_t W, uint32_t H, uint32_t world) { return (tile_count(W, H) + world - 1) / world; }
inline size_t packed_floats(uint32_t W, uint32_t H, uint32_t world) { return (size_t)slot_count(W, H, world) * kSlotPixels * kPixelFloats; }

struct slot_px { uint32_t x, y; bool inside, halo, foreign; };   // frame pixel of position (lx, ly) of slot k of `rank`; foreign: owned by another rank and carried by THIS slot's halo
__device__ __forceinline__ slot_px slot_pixel(uint32_t k, uint32_t lx, uint32_t ly, uint32_t W, uint32_t H, uint32_t rank, uint32_t world) {
    const uint32_t tiles_x = (W + 63u) >> 6, tile = k * world + rank;
    slot_px q; q.x = (tile % tiles_x) * 64u + lx; q.y = (tile / tiles_x) * 64u + ly;
    q.halo = lx == 64u || ly == 64u;
    q.inside = tile < tile_count(W, H) && q.x < W && q.y < H;   // (x < W: the halo of a tile at the right edge of the film does not wrap into the next row)
    q.foreign = q.inside && (((q.y >> 6) * tiles_x + (q.x >> 6)) % world) != rank;
    // A foreign pixel at the corner of its tile lies in the halo of up to three tiles (the one on its left, the one above, the diagonal one) and the rank may own more than
    // one of them: it travels with exactly one — left before above before diagonal.  (inside => those tiles exist.)
    if (q.foreign && ly == 64u) {
        if (lx == 0u && tile % tiles_x != 0u && (tile + tiles_x - 1u) % world == rank) q.foreign = false;                       // bottom row, first pixel: the left tile of that pixel is ours too
        if (lx == 64u && ((tile + tiles_x) % world == rank || (tile + 1u) % world == rank)) q.foreign = false;                  // corner: its left tile (below us) or its upper tile (right of us) is ours
    }
    return q;
}
// frame -> the rank's slots.  n = packed_floats
__global__ __launch_bounds__(256) void k_pack_tiles(const float* __restrict__ frame, float* __restrict__ packed, size_t n, uint32_t W, uint32_t H, uint32_t rank, uint32_t world) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t px = (uint32_t)(i / kPixelFloats), c = (uint32_t)(i % kPixelFloats), k = px / kSlotPixels, p = px % kSlotPixels;
    const slot_px q = slot_pixel(k, p % kSlotEdge, p / kSlotEdge, W, H, rank, world);
    packed[i] = (q.inside && (!q.halo || q.foreign)) ? frame[((size_t)q.y * W + q.x) * kPixelFloats + c] : 0.0f;
}
// phase 1 on the root: the 64 x 64 interiors of every rank's slots -> frame (a copy; blockIdx.y = the rank whose slots these are, or `rank0` when gridDim.y == 1)
__global__ __launch_bounds__(256) void k_unpack_tiles(float* __restrict__ frame, const float* __restrict__ packed, size_t n, uint32_t W, uint32_t H, uint32_t rank0, uint32_t world) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rank0 + blockIdx.y;
    const uint32_t px = (uint32_t)(i / kPixelFloats), c = (uint32_t)(i % kPixelFloats), k = px / kSlotPixels, p = px % kSlotPixels;
    const slot_px q = slot_pixel(k, p % kSlotEdge, p / kSlotEdge, W, H, r, world);
    if (q.inside && !q.halo) frame[((size_t)q.y * W + q.x) * kPixelFloats + c] = packed[(size_t)blockIdx.y * n + i];
}
// phase 2, after phase 1 of ALL ranks: the halos are added (float atomics: a pixel can receive from the tile on its left, above it and diagonally; two such samples in one
// pixel of one render are a ~10^-8 event, and their order is the only thing that is not fixed).  One lane per halo float: h < 65 = the bottom row, else the right column.
__global__ __launch_bounds__(256) void k_add_halos(float* __restrict__ frame, const float* __restrict__ packed, size_t n_slot_floats, uint32_t n_slots, uint32_t W, uint32_t H, uint32_t rank0, uint32_t world) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_slots * kHaloPixels * kPixelFloats) return;
    const uint32_t r = rank0 + blockIdx.y;
    const uint32_t hp = (uint32_t)(i / kPixelFloats), c = (uint32_t)(i % kPixelFloats), k = hp / kHaloPixels, h = hp % kHaloPixels;
    const uint32_t lx = h < kSlotEdge ? h : 64u, ly = h < kSlotEdge ? 64u : h - kSlotEdge;
    const slot_px q = slot_pixel(k, lx, ly, W, H, r, world);
    if (!q.foreign) return;
    const float v = packed[(size_t)blockIdx.y * n_slot_floats + ((size_t)k * kSlotPixels + ly * kSlotEdge + lx) * kPixelFloats + c];
    if (v != 0.0f) atomicAdd(&frame[((size_t)q.y * W + q.x) * kPixelFloats + c], v);
}
void launch_pack(hipStream_t s, Image* img, float* packed, uint32_t rank, uint32_t world) {
    const size_t n = packed_floats(img->getWidth(), img->getHeight(), world);
    hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)img->device(), packed, n, img->getWidth(), img->getHeight(), rank, world);
}
// packed: the slots of ranks rank0 .. rank0 + n_ranks - 1, rank-major
void launch_unpack(hipStream_t s, Image* img, const float* packed, uint32_t rank0, uint32_t n_ranks, uint32_t world) {
    const uint32_t W = img->getWidth(), H = img->getHeight(), slots = slot_count(W, H, world);
    const size_t n = packed_floats(W, H, world), nh = (size_t)slots * kHaloPixels * kPixelFloats;
    hipLaunchKernelGGL(k_unpack_tiles, dim3((unsigned)((n + 255) / 256), n_ranks), dim3(256), 0, s, (float*)img->device(), packed, n, W, H, rank0, world);
    hipLaunchKernelGGL(k_add_halos, dim3((unsigned)((nh + 255) / 256), n_ranks), dim3(256), 0, s, (float*)img->device(), packed, n, slots, W, H, rank0, world);
}
void check_shard(uint32_t rank, uint32_t world) { if (world == 0 || rank >= world) throw std::runtime_error("bad tile shard (rank / world)"); }
}  // namespace

size_t image_packed_tile_bytes(uint32_t W, uint32_t H, uint32_t world) { if (world == 0) throw std::runtime_error("bad tile shard (world = 0)"); return packed_floats(W, H, world) * sizeof(float); }
// host-side exchange (a host without RCCL: MPI_Gather of these buffers; bench.py's gloo fallback): the same kernels, staged through a device buffer of the call
void image_pack_tiles(Image* img, uint32_t rank, uint32_t world, void* host_out) {
    check_shard(rank, world); require_device();
    dbuf<float> tmp; tmp.alloc(packed_floats(img->getWidth(), img->getHeight(), world));
    CTL_HIP(hipDeviceSynchronize());   // the frame is complete (the tracer renders on its own stream)
    launch_pack(nullptr, img, tmp.p, rank, world);
    CTL_HIP(hipMemcpy(host_out, tmp.p, tmp.n * sizeof(float), hipMemcpyDeviceToHost));
}
// host_in: the buffers of ALL `world` ranks, rank-major (what MPI_Gather delivers on the root): every tile is copied, then every halo added
void image_unpack_tiles(Image* img, uint32_t world, const void* host_in) {
    check_shard(0, world); require_device();
    dbuf<float> tmp; tmp.alloc(packed_floats(img->getWidth(), img->getHeight(), world) * world);
    CTL_HIP(hipMemcpy(tmp.p, host_in, tmp.n * sizeof(float), hipMemcpyHostToDevice));
    CTL_HIP(hipDeviceSynchronize());
    launch_unpack(nullptr, img, tmp.p, 0u, world, world);
    CTL_HIP(hipDeviceSynchronize());
    if (world > 1) img->mark_reduced(true);   // holds the other ranks' tiles now: an in-place exchange of it would count them twice
}

struct Comm {
    ncclComm_t comm = nullptr; int rank = 0, world = 1; hipStream_t stream = nullptr; hipEvent_t done = nullptr;
    int timeout_ms = 0;        // of every collective on this communicator (ctl_comm_create_timeout's argument)
    bool dead = false;         // a collective timed out and the communicator was aborted: every later call is refused, the destructor does not enter ncclCommDestroy
    bool stuck = false;        // ... and the abort was not available or failed: a collective may still be reading / writing send / recv and sits on `stream` — those three are LEAKED, never freed or synchronised
    dbuf<float> send, recv;    // packed tiles of this rank / of every rank (root only), kept between gathers
    ~Comm() {
        if (comm && !dead) (void)rccl().CommDestroy(comm);
        if (stuck) { send.p = nullptr; send.n = 0; recv.p = nullptr; recv.n = 0; return; }   // hipFree / hipStreamDestroy / hipEventDestroy would wait for the stuck collective
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
// how long a collective call may take before it is given up (ms; ctl_comm_create_timeout's argument, else $CTL_COMM_TIMEOUT_MS, else 120 s).  A rank that never arrives makes
// ncclCommInitRank / ncclReduce / ncclGather wait for ever: the caller gets an error it can act on (bench.py falls back to torch.distributed on every rank) instead of a hung job.
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
    // ncclCommInitRank is collective and has no deadline of its own: it runs on a helper thread (on this thread's device) and is waited for.  When the wait is given up
    // the thread is told so: a communicator that arrives after the deadline is aborted by the thread itself instead of leaking.
    struct pending { std::promise<std::pair<ncclResult_t, ncclComm_t>> p; std::mutex m; bool abandoned = false; };
    auto st = std::make_shared<pending>();
    auto fut = st->p.get_future();
    std::thread([st, &api, id, rank, world, dev]() {
        ncclComm_t c = nullptr; ncclResult_t r = ncclSystemError;
        if (hipSetDevice(dev) == hipSuccess) r = api.CommInitRank(&c, world, id, rank);   // one rank per process, on the process's current device
        std::lock_guard<std::mutex> g(st->m);
        if (st->abandoned) { if (r == ncclSuccess && c) (void)(api.CommAbort ? api.CommAbort(c) : api.CommDestroy(c)); return; }
        st->p.set_value({ r, c });
    }).detach();
    if (fut.wait_for(std::chrono::milliseconds(timeout_ms)) != std::future_status::ready) {
        std::unique_lock<std::mutex> g(st->m);
        if (fut.wait_for(std::chrono::milliseconds(0)) != std::future_status::ready) {   // (not set between the wait and the lock)
            st->abandoned = true;
            throw std::runtime_error("ctl_comm_create: ncclCommInitRank of rank " + std::to_string(rank) + " / " + std::to_string(world) + " did not return within " + std::to_string(timeout_ms) + " ms (a rank that never arrived?)");
        }
    }
    const auto res = fut.get();
    check(res.first, "ncclCommInitRank");
    std::unique_ptr<Comm> c(new Comm());
    c->rank = rank; c->world = world; c->comm = res.second; c->timeout_ms = timeout_ms;
    CTL_HIP(hipStreamCreate(&c->stream));
    CTL_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    return c.release();
}
void comm_destroy(Comm* c) { delete c; }

// everything queued on c->stream is complete, or — after the communicator's time-out — the communicator is aborted (so that no collective is left reading or writing the
// caller's buffers when the caller frees them), marked dead, and the call throws
static void wait_done(Comm* c, const char* what) {
    CTL_HIP(hipEventRecord(c->done, c->stream));
    const auto start = std::chrono::steady_clock::now(), deadline = start + std::chrono::milliseconds(c->timeout_ms);
    for (;;) {
        const hipError_t q = hipEventQuery(c->done);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) CTL_HIP(q);
        const auto now = std::chrono::steady_clock::now();
        if (now > deadline) {
            c->dead = true;
            // ncclCommAbort ends the kernel of the collective and the stream drains: only THEN is it safe to wait for the stream.  Without the symbol (the dlsym is optional) or
            // when the abort fails, waiting would block for ever on the very collective the deadline is for: throw at once and leak the buffers it may still touch (Comm::stuck)
            const bool aborted = rccl().CommAbort && rccl().CommAbort(c->comm) == ncclSuccess;
            if (aborted) (void)hipStreamSynchronize(c->stream); else c->stuck = true;
            throw std::runtime_error(std::string(what) + " did not complete within " + std::to_string(c->timeout_ms) + " ms; " +
                                     (aborted ? "the communicator was aborted" : "the communicator could not be aborted (its buffers and stream are left to the stuck collective)") + " (create a new one, or fall back)");
        }
        if (now - start > std::chrono::milliseconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // spin for the first 5 ms (an exchange takes < 1), then poll
    }
}
static void require_alive(Comm* c, const char* what) { if (c->dead) throw std::runtime_error(std::string(what) + ": this communicator was aborted after a time-out"); }

// sum over the ranks of `src` (every rank's own PixelData frame) into `dst` on the root; dst == src is the in-place form.  dst is only read on the root (may be null elsewhere).
// Returns when the reduce is complete, or throws when it did not complete within the time-out.
void comm_reduce_image(Comm* c, Image* src, Image* dst, int root) {
    require_alive(c, "ctl_image_reduce");
    if (root < 0 || root >= c->world) throw std::runtime_error("ctl_image_reduce: bad root");
    if (c->rank == root && !dst) throw std::runtime_error("ctl_image_reduce_to: the root needs a destination image");
    if (dst && (dst->getWidth() != src->getWidth() || dst->getHeight() != src->getHeight())) throw std::runtime_error("ctl_image_reduce_to: source and destination sizes differ");
    // the in-place form is marked on EVERY rank (below), so that a repeated call is refused by all ranks together, before any of them enters the collective; with one
    // rank the sum is the frame itself and a repeat is harmless
    if (c->world > 1 && dst == src && src->holds_reduced_frame())
        throw std::runtime_error("ctl_image_reduce: this image already went through an in-place reduce — a second one would add the other ranks' tiles onto sums that contain them; "
                                 "use ctl_image_gather, or ctl_image_reduce_to for a per-pass (progressive) exchange, or clear the image first");
    CTL_HIP(hipDeviceSynchronize());   // the frame is complete (the tracer renders on its own stream)
    const size_t n = (size_t)src->getWidth() * src->getHeight() * kPixelFloats;
    check(rccl().Reduce(src->device(), dst ? dst->device() : src->device(), n, ncclFloat, ncclSum, root, c->comm, c->stream), "ncclReduce");
    wait_done(c, "ctl_image_reduce: ncclReduce");
    if (dst == src && c->world > 1) src->mark_reduced(true);
}

// the root's `dst` receives every rank's own tiles of `src` (dst == src: in place; dst is only used on the root).  ONE ncclGather of ceil(tiles / world) packed tiles per rank.
void comm_gather_image(Comm* c, Image* src, Image* dst, int root) {
    require_alive(c, "ctl_image_gather");
    if (root < 0 || root >= c->world) throw std::runtime_error("ctl_image_gather: bad root");
    if (c->rank == root && !dst) throw std::runtime_error("ctl_image_gather_to: the root needs a destination image");
    if (dst && (dst->getWidth() != src->getWidth() || dst->getHeight() != src->getHeight())) throw std::runtime_error("ctl_image_gather_to: source and destination sizes differ");
    // With more than one rank the in-place form is ONE call per render, like the reduce: the root's own tiles have received the other ranks' halo samples, and packing them
    // again would carry those along while the halos are added once more.  Marked and refused on every rank together.  (gather_to leaves src alone and can be repeated.)
    if (c->world > 1 && dst == src && src->holds_reduced_frame())
        throw std::runtime_error("ctl_image_gather: this image already went through an in-place exchange; use ctl_image_gather_to for a per-pass (progressive) gather, or clear the image first");
    const rccl_api& api = rccl();
    if (!api.Gather) throw std::runtime_error("ctl_image_gather: this librccl.so has no ncclGather (use ctl_image_reduce)");
    const uint32_t W = src->getWidth(), H = src->getHeight();
    const size_t n = packed_floats(W, H, (uint32_t)c->world);
    if (c->send.n != n) c->send.alloc(n);
    if (c->rank == root && c->recv.n != n * c->world) c->recv.alloc(n * c->world);
    CTL_HIP(hipDeviceSynchronize());   // the frame is complete (the tracer renders on its own stream)
    launch_pack(c->stream, src, c->send.p, (uint32_t)c->rank, (uint32_t)c->world);
    check(api.Gather(c->send.p, c->rank == root ? c->recv.p : nullptr, n, ncclFloat, root, c->comm, c->stream), "ncclGather");
    if (c->rank == root) launch_unpack(c->stream, dst, c->recv.p, 0u, (uint32_t)c->world, (uint32_t)c->world);   // every tile copied, then every halo added
    wait_done(c, "ctl_image_gather: ncclGather");
    // a gathered frame holds every rank's tiles: an in-place exchange of it would count them twice — marked on every rank like the reduce's own mark
    if (c->world > 1) { if (dst == src) src->mark_reduced(true); else if (c->rank == root) dst->mark_reduced(true); }
}
size_t comm_gather_bytes(Comm* c, uint32_t W, uint32_t H) { return packed_floats(W, H, (uint32_t)c->world) * sizeof(float); }

}  // namespace ctl
