// pt_multi.hpp — a render context made of several wavefront streams, behind the same C ABI (included by pt_hip.hip).
//
// Two uses of one mechanism:
//  * several GPUs.  The reference is ONE host thread that owns ONE GL context and issues one draw call per frame
//    (/root/reference/src/Main/dispatch.java:168, :593-713).  A caller of that shape (the Java Main through the JNI shim, a C
//    program) gets all GPUs of the node through ONE context: pt_create_multi(devices[], n, W, H) makes a group whose entry points
//    are the ordinary ones — pt_set_buffer replicates the scene, pt_render / pt_render_batch(_async) render every stream's tile
//    shard (32x8 tiles dealt round-robin), pt_read_frame / pt_gather_image perform the ONE collective per image:
//    an RCCL gather (ncclGather over xGMI, single process, ncclCommInitAll) of the packed shard accumulators on devices[0] and the
//    un-tiling kernel there.  No exchange during rendering; results are bit-identical for every stream count (K10).
//  * several streams on ONE GPU: a device listed k times carries k shards, each with its own path pool and HIP stream.  The
//    streams run unsynchronised, so the intersect kernel of one (instruction-issue and latency bound) overlaps the shading kernel of
//    another (HBM bound) and every launch's tail is filled by the other stream's blocks: two streams per GPU render C3 12 %, C4
//    25 % faster than one (profiles/r02_g_streams_per_gpu.txt, r02_y_block_size.txt).  Shards that share a device are gathered with device copies.
//  The two combine ({0,0,1,1,...}: equal multiplicity, a device's entries adjacent): per device its shards are copied into one
//  staging block, the blocks travel in the one ncclGather.  pt_create_multi_part makes a group that holds only shards
//  first..first+n-1 of a larger total (one process per GPU, each with its own streams): its gather stops at the packed block.
//
// Host side: every stream's context is driven by its own host thread (the wavefront scheduler polls its device, pump()), so the
// schedulers run concurrently; the group's entry points hand the call to the threads and join them.
#pragma once
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

#include <rccl/rccl.h>

namespace {

struct Rccl {                       // librccl is loaded when the first multi-device gather needs it: one-GPU contexts never touch it
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    int load() {
        if (lib) return 0;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return fail(PT_ERR_HIP, std::string("multi-GPU gather needs RCCL: ") + dlerror());
#define RCCL_SYM(f) do { f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) return fail(PT_ERR_HIP, "librccl lacks nccl" #f); } while (0)
        RCCL_SYM(CommInitAll); RCCL_SYM(CommDestroy); RCCL_SYM(GroupStart); RCCL_SYM(GroupEnd); RCCL_SYM(Gather); RCCL_SYM(GetErrorString);
#undef RCCL_SYM
        return 0;
    }
};
Rccl g_rccl;
#define RCCL_TRY(x)                                                                                                    \
    do {                                                                                                               \
        ncclResult_t r_ = (x);                                                                                         \
        if (r_ != ncclSuccess) return fail(PT_ERR_HIP, std::string(#x) + ": " + g_rccl.GetErrorString(r_));             \
    } while (0)

struct Worker {                     // one host thread per device context
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool pending = false, quit = false;
    int rc = 0;
    std::string err;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return pending || quit; });
            if (quit) return;
            lk.unlock();
            int r = job();
            std::string e = r ? g_err : std::string();
            lk.lock();
            rc = r; err = e; pending = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f) { std::lock_guard<std::mutex> lk(m); job = std::move(f); pending = true; cv.notify_all(); }
    int wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !pending; }); return rc; }
};

}  // namespace

struct MultiCtx {
    int n = 0;                          // streams (shards held by this group)
    int shardBase = 0, shardTotal = 0;  // they are shards shardBase .. shardBase+n-1 of shardTotal (== n: the whole image)
    std::vector<int> devices;           // per stream; equal entries adjacent
    std::vector<pt_ctx*> kids;
    struct Run { int device, first, count; float4* staging; };      // the streams of one device
    std::vector<Run> runs;
    bool useRccl = false;               // more than one distinct device (or PT_MULTI_FORCE_RCCL=1: the RCCL call path on a one-GPU box)
    bool virtualDevices = false;        // test mode PT_MULTI_VIRTUAL_DEVICES: the runs share one GPU, the transport between them is device copies
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<ncclComm_t> comms;      // one per run
    std::vector<hipEvent_t> ev;         // per stream: "this shard's image is complete"; ev[run.first] doubles as "the copies have read it"
    float4* dGathered = nullptr;        // on devices[0]: n * nSlots packed accumulators, shard-major (what the gather delivers)
    float4* dFull = nullptr;            // on devices[0]: W * H, un-tiled (whole-image groups)
    int* dAllMaps = nullptr;            // on devices[0]: packed slot -> global pixel, -1 = padding
    std::vector<int32_t> maps;          // the same on the host (pt_read_frame of a partial group)
    bool gatherReady = false;           // the buffers above exist (allocated by the first gather)
    std::set<int> staleBindings;        // bindings (textures: 1000 + index) whose last upload reached only some of the streams: no render until repeated
    uint64_t gathers = 0;
};

namespace {

int multiRun(MultiCtx& M, const std::function<int(pt_ctx*)>& f) {
    for (int i = 0; i < M.n; i++) { pt_ctx* k = M.kids[i]; M.workers[i]->post([&f, k] { return f(k); }); }
    int rc = 0; std::string err;
    for (int i = 0; i < M.n; i++) {
        int r = M.workers[i]->wait();
        if (r && !rc) { rc = r; err = "device " + std::to_string(M.devices[i]) + " (shard " + std::to_string(i) + "): " + M.workers[i]->err; }
    }
    return rc ? fail(rc, err) : 0;
}

void multiFree(pt_ctx* g) {
    MultiCtx* M = g->multi;
    if (!M) return;
    // Order: nothing in flight on any device first; then the communicators, while the streams their collectives ran on still exist (RCCL keeps
    // references to the user streams of its last operations); then the streams' contexts; the gather buffers last.
    for (int d : std::set<int>(M->devices.begin(), M->devices.end())) { hipSetDevice(d); hipDeviceSynchronize(); }
    if (!M->devices.empty()) hipSetDevice(M->devices[0]);
    for (ncclComm_t& c : M->comms) if (c && g_rccl.CommDestroy) { g_rccl.CommDestroy(c); c = nullptr; }
    for (int i = 0; i < (int)M->kids.size(); i++) {
        pt_ctx* k = M->kids[i];
        if (i < (int)M->workers.size() && M->workers[i]) { M->workers[i]->post([k] { return pt_destroy(k); }); M->workers[i]->wait(); }
        else pt_destroy(k);
    }
    for (auto& w : M->workers) if (w) { { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); } if (w->th.joinable()) w->th.join(); }
    if (!M->devices.empty()) hipSetDevice(M->devices[0]);
    for (hipEvent_t e : M->ev) if (e) hipEventDestroy(e);
    for (void* p : {(void*)M->dGathered, (void*)M->dFull, (void*)M->dAllMaps}) if (p) hipFree(p);
    for (auto& r : M->runs) if (r.staging) { hipSetDevice(r.device); hipFree(r.staging); }
    delete M;
    g->multi = nullptr;
}

// The ONE collective of an image: every stream completes image `age`; the shards of one device are copied into one block there, the
// blocks go to devices[0] (ncclGather over xGMI when there are several devices), and — for a group that holds the whole image —
// the un-tiling kernel writes the W x H image.  Stream-ordered on the root stream (kids[0]'s), not synchronised.
// *out: the whole image (shardTotal == n) or the group's packed block of n * nSlots accumulators, shard-major.
int multiGather(pt_ctx* g, int age, float4** out) {
    MultiCtx& M = *g->multi;
    int rc;
    if ((rc = multiRun(M, [age](pt_ctx* k) { return pt_finish_image(k, age); }))) return rc;
    pt_ctx* root = M.kids[0];
    const size_t nSlots = (size_t)root->nSlotsImg, total = nSlots * (size_t)M.n;
    std::vector<float4*> img(M.n);
    for (int i = 0; i < M.n; i++) {
        pt_ctx* k = M.kids[i];
        img[i] = k->dImage[(k->curImage + pt_ctx::IMAGES - age) % pt_ctx::IMAGES];
        if (!img[i]) return fail(PT_ERR_ARG, "no image of that age yet (too few pt_next_image calls)");
    }
    HIP_TRY(hipSetDevice(M.devices[0]));
    if (!M.gatherReady) {                  // first gather: its buffers.  A failure half-way leaves gatherReady false: the next call starts over
        for (float4** p : {&M.dGathered, &M.dFull}) if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; }
        if (M.dAllMaps) { HIP_TRY(hipFree(M.dAllMaps)); M.dAllMaps = nullptr; }
        HIP_TRY(hipMalloc((void**)&M.dGathered, total * 16));
        M.maps.resize(total);
        for (int r = 0; r < M.n; r++) if ((rc = pt_shard_map(g->W, g->H, M.shardBase + r, M.shardTotal, M.maps.data() + (size_t)r * nSlots, nSlots))) return rc;
        if (M.shardTotal == M.n) {
            HIP_TRY(hipMalloc((void**)&M.dFull, (size_t)g->W * g->H * 16));
            HIP_TRY(hipMalloc((void**)&M.dAllMaps, total * 4));
            HIP_TRY(hipMemcpy(M.dAllMaps, M.maps.data(), total * 4, hipMemcpyHostToDevice));
        }
        if (M.ev.empty()) M.ev.assign(M.n, nullptr);
        for (int i = 0; i < M.n; i++) if (!M.ev[i]) { HIP_TRY(hipSetDevice(M.devices[i])); HIP_TRY(hipEventCreateWithFlags(&M.ev[i], hipEventDisableTiming)); }
        HIP_TRY(hipSetDevice(M.devices[0]));
        M.gatherReady = true;
    }
    // 1. per device: its streams' accumulators side by side, on the device's first stream (one stream: nothing to copy).  Without RCCL
    //    (one device) the block IS the gathered buffer.
    for (auto& run : M.runs) {
        if (run.count == 1 && M.useRccl) continue;
        HIP_TRY(hipSetDevice(run.device));
        pt_ctx* lead = M.kids[run.first];
        float4* dst = M.useRccl ? run.staging : M.dGathered + (size_t)run.first * nSlots;
        if (M.useRccl && !dst) { HIP_TRY(hipMalloc((void**)&run.staging, (size_t)run.count * nSlots * 16)); dst = run.staging; }
        for (int j = 0; j < run.count; j++) {
            const int i = run.first + j;
            if (j > 0) { HIP_TRY(hipEventRecord(M.ev[i], M.kids[i]->stream)); HIP_TRY(hipStreamWaitEvent(lead->stream, M.ev[i], 0)); }
            HIP_TRY(hipMemcpyAsync(dst + (size_t)j * nSlots, img[i], nSlots * 16, hipMemcpyDeviceToDevice, lead->stream));
        }
        // the other streams must not rotate their image rings past this image before the copies have read it
        if (run.count > 1) {
            HIP_TRY(hipEventRecord(M.ev[run.first], lead->stream));
            for (int j = 1; j < run.count; j++) HIP_TRY(hipStreamWaitEvent(M.kids[run.first + j]->stream, M.ev[run.first], 0));
        }
    }
    // 2. across devices: ONE ncclGather, a block per device, root = devices[0]
    if (M.useRccl) {
        if (!M.virtualDevices && (rc = g_rccl.load())) return rc;
        if (!M.virtualDevices && M.comms.empty()) {
            std::vector<int> devs;
            for (auto& run : M.runs) devs.push_back(run.device);
            M.comms.assign(M.runs.size(), nullptr);
            const ncclResult_t e = g_rccl.CommInitAll(M.comms.data(), (int)devs.size(), devs.data());
            if (e != ncclSuccess) { M.comms.clear(); return fail(PT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e)); }      // the next call starts over
        }
        if (!M.virtualDevices) RCCL_TRY(g_rccl.GroupStart());
        for (size_t r = 0; r < M.runs.size(); r++) {
            const auto& run = M.runs[r];
            // every exit from inside the group closes it: an open group would swallow the next call's collectives
            { const hipError_t he = hipSetDevice(run.device); if (he != hipSuccess) { if (!M.virtualDevices) g_rccl.GroupEnd(); return fail(PT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he)); } }
            const float4* send = run.count == 1 ? img[run.first] : run.staging;
            // recvbuff matters on the root only; the other ranks pass a valid local pointer that is never written
            void* const recv = r == 0 ? (void*)M.dGathered : (void*)send;
            const size_t floats = (size_t)run.count * nSlots * 4;
            hipStream_t const strm = M.kids[run.first]->stream;
            if (M.virtualDevices) {
                // what ncclGather does with these arguments, by device copies: rank r's block lands at r * count in the root's receive buffer,
                // and the root's stream continues once every block has arrived
                hipError_t he = hipMemcpyAsync((float*)M.dGathered + r * floats, send, floats * 4, hipMemcpyDeviceToDevice, strm);
                if (he == hipSuccess && r > 0) { he = hipEventRecord(M.ev[run.first], strm); if (he == hipSuccess) he = hipStreamWaitEvent(root->stream, M.ev[run.first], 0); }
                if (he != hipSuccess) return fail(PT_ERR_HIP, std::string("virtual-device gather: ") + hipGetErrorString(he));
                (void)recv;
                continue;
            }
            ncclResult_t e = g_rccl.Gather(send, recv, floats, ncclFloat, 0, M.comms[r], strm);
            if (e != ncclSuccess) { g_rccl.GroupEnd(); return fail(PT_ERR_HIP, std::string("ncclGather: ") + g_rccl.GetErrorString(e)); }
        }
        if (!M.virtualDevices) RCCL_TRY(g_rccl.GroupEnd());
    }
    HIP_TRY(hipSetDevice(M.devices[0]));
    M.gathers++;
    if (M.shardTotal != M.n) { *out = M.dGathered; return 0; }     // a part of the image: the host layer gathers the blocks of all processes
    hipLaunchKernelGGL(k_unshard, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, root->stream, (const float4*)M.dGathered, M.dAllMaps, (int)nSlots, M.n, M.dFull);
    HIP_TRY(hipGetLastError());
    *out = M.dFull;
    return 0;
}

}  // namespace
