// pt_multi.hpp — the multi-GPU form of a render context, behind the same C ABI (included by pt_hip.hip).
//
// The reference is ONE host thread that owns ONE GL context and issues one draw call per frame
// (/root/reference/src/Main/dispatch.java:168, :593-713).  A caller of that shape (the Java Main through the JNI shim, a C
// program) gets all GPUs of the node through ONE context: pt_create_multi(devices[], n, W, H) makes a group whose entry points
// are the ordinary ones — pt_set_buffer replicates the scene, pt_render / pt_render_batch(_async) render every device's tile
// shard (32x8 tiles dealt round-robin), pt_read_frame / pt_gather_image perform the ONE collective per image:
// an RCCL gather (ncclGather over xGMI, single process, ncclCommInitAll) of the packed shard accumulators on device[0] and the
// un-tiling kernel there.  No exchange during rendering; results are bit-identical for every device count (K10).
//
// Host side: each device's context is driven by its own host thread (the wavefront scheduler polls its device, pump()), so the N
// schedulers run concurrently; the group's entry points hand the call to the N threads and join them.
//
// A device listed more than once ({0,0}: two shards on one GPU) is a rehearsal of the sharding on fewer GPUs than shards — RCCL
// refuses two ranks on one device — and gathers with device-to-device copies instead; the bookkeeping (shard maps, padding,
// un-tiling, image ring) is the same code.
#pragma once
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <mutex>
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
    int n = 0;
    std::vector<int> devices;
    std::vector<pt_ctx*> kids;
    bool sameDevice = false;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<ncclComm_t> comms;
    std::vector<hipEvent_t> ev;         // copy path: "this shard's image is complete" on the shard's stream
    float4* dGathered = nullptr;        // on devices[0]: n * nSlots packed accumulators, rank-major (what ncclGather delivers)
    float4* dFull = nullptr;            // on devices[0]: W * H, un-tiled
    int* dAllMaps = nullptr;            // on devices[0]: packed slot -> global pixel, -1 = padding
    unsigned char* dDisplay = nullptr;
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
    for (int i = 0; i < (int)M->kids.size(); i++) {
        pt_ctx* k = M->kids[i];
        if (i < (int)M->workers.size() && M->workers[i]) { M->workers[i]->post([k] { return pt_destroy(k); }); M->workers[i]->wait(); }
        else pt_destroy(k);
    }
    for (auto& w : M->workers) if (w) { { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); } if (w->th.joinable()) w->th.join(); }
    if (!M->devices.empty()) hipSetDevice(M->devices[0]);
    for (ncclComm_t c : M->comms) if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
    for (hipEvent_t e : M->ev) if (e) hipEventDestroy(e);
    for (void* p : {(void*)M->dGathered, (void*)M->dFull, (void*)M->dAllMaps, (void*)M->dDisplay}) if (p) hipFree(p);
    delete M;
    g->multi = nullptr;
}

// The ONE collective of an image: every shard completes image `age`, its packed accumulator goes to devices[0] (RCCL gather over
// xGMI), and the un-tiling kernel writes the full W x H image there.  Stream-ordered on the root shard's stream.
int multiGather(pt_ctx* g, int age, float4** fullOut) {
    MultiCtx& M = *g->multi;
    int rc;
    if ((rc = multiRun(M, [age](pt_ctx* k) { return pt_finish_image(k, age); }))) return rc;
    pt_ctx* root = M.kids[0];
    HIP_TRY(hipSetDevice(M.devices[0]));
    const size_t nSlots = (size_t)root->nSlotsImg, total = nSlots * (size_t)M.n;
    std::vector<float4*> img(M.n);
    for (int i = 0; i < M.n; i++) {
        pt_ctx* k = M.kids[i];
        img[i] = k->dImage[(k->curImage + pt_ctx::IMAGES - age) % pt_ctx::IMAGES];
        if (!img[i]) return fail(PT_ERR_ARG, "no image of that age yet (too few pt_next_image calls)");
    }
    if (!M.dGathered) {
        HIP_TRY(hipMalloc((void**)&M.dGathered, total * 16));
        HIP_TRY(hipMalloc((void**)&M.dFull, (size_t)g->W * g->H * 16));
        std::vector<int32_t> maps(total);
        for (int r = 0; r < M.n; r++) if ((rc = pt_shard_map(g->W, g->H, r, M.n, maps.data() + (size_t)r * nSlots, nSlots))) return rc;
        HIP_TRY(hipMalloc((void**)&M.dAllMaps, total * 4));
        HIP_TRY(hipMemcpy(M.dAllMaps, maps.data(), total * 4, hipMemcpyHostToDevice));
    }
    if (!M.sameDevice) {
        if ((rc = g_rccl.load())) return rc;
        if (M.comms.empty()) {
            M.comms.assign(M.n, nullptr);
            RCCL_TRY(g_rccl.CommInitAll(M.comms.data(), M.n, M.devices.data()));
        }
        RCCL_TRY(g_rccl.GroupStart());
        for (int i = 0; i < M.n; i++) {
            HIP_TRY(hipSetDevice(M.devices[i]));
            // recvbuff matters on the root only; the other ranks pass a valid local pointer that is never written
            ncclResult_t r = g_rccl.Gather(img[i], i == 0 ? (void*)M.dGathered : (void*)img[i], nSlots * 4, ncclFloat, 0, M.comms[i], M.kids[i]->stream);
            if (r != ncclSuccess) { g_rccl.GroupEnd(); return fail(PT_ERR_HIP, std::string("ncclGather: ") + g_rccl.GetErrorString(r)); }
        }
        RCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(hipSetDevice(M.devices[0]));
    } else {
        if (M.ev.empty()) { M.ev.assign(M.n, nullptr); for (int i = 0; i < M.n; i++) { HIP_TRY(hipSetDevice(M.devices[i])); HIP_TRY(hipEventCreateWithFlags(&M.ev[i], hipEventDisableTiming)); } }
        for (int i = 0; i < M.n; i++) {
            HIP_TRY(hipSetDevice(M.devices[i]));
            HIP_TRY(hipEventRecord(M.ev[i], M.kids[i]->stream));
        }
        HIP_TRY(hipSetDevice(M.devices[0]));
        for (int i = 0; i < M.n; i++) {
            HIP_TRY(hipStreamWaitEvent(root->stream, M.ev[i], 0));
            HIP_TRY(hipMemcpyAsync(M.dGathered + (size_t)i * nSlots, img[i], nSlots * 16, hipMemcpyDeviceToDevice, root->stream));
        }
        // the shards must not rotate their image rings past this image before the copies have read it
        HIP_TRY(hipEventRecord(M.ev[0], root->stream));
        for (int i = 1; i < M.n; i++) { HIP_TRY(hipSetDevice(M.devices[i])); HIP_TRY(hipStreamWaitEvent(M.kids[i]->stream, M.ev[0], 0)); }
        HIP_TRY(hipSetDevice(M.devices[0]));
    }
    hipLaunchKernelGGL(k_unshard, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, root->stream, (const float4*)M.dGathered, M.dAllMaps, (int)nSlots, M.n, M.dFull);
    HIP_TRY(hipGetLastError());
    M.gathers++;
    *fullOut = M.dFull;
    return 0;
}

}  // namespace
