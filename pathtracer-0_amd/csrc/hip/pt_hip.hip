// pt_hip.hip — wavefront path tracer for MI355X (gfx950) behind the C ABI of include/pt_api.h.
//
// Restates the per-pixel Monte-Carlo loop of the reference's fragment shader
// (/root/reference/src/shaders/frag.glsl:884-934) as a wavefront state machine:
//
//   k_frame_setup   uniform work hoisted out of the per-pixel loop: rotationMatrix(ROTATION)
//                   (:271-283) and the auto-focus ray (:901-906, identical for every pixel)
//   k_revive        starts jobs in dead path slots: job -> pixel, rngState = index + u_seed (:896), first camera ray
//                   (:899-908), trace() prologue (:811-818); slot i takes job i when a frame stream starts
//   k_extend_persist  rayScene (:548-653): per-object BVH traversal + ellipsoids; persistent blocks, LDS-staged top of
//                   the BVH, in-wave ray refill, one instruction stream per trip chosen by vote   [the hot kernel]
//                   (k_extend: the simple one-block-per-256-rays form, kept as the cross-check)
//   k_shade         trace() loop body (:823-879): material, index stack, chooseRay, absorption,
//                   emission, cut-off, throughput; sky on miss; sample end -> regenerate the next
//                   sample of the pixel-frame in place (serial rngState) or pull a
//                   new job with one block-aggregated atomic; finished pixel-frames go to `colbuf`;
//                   once job supply runs dry it packs the surviving slots into the next iteration's queue
//   k_scan_inflight is a live slot still on a frame of the oldest unaccumulated batch? (one pass per host poll)
//   k_accumulate    FRAME accumulation (:924-933) of one batch in u_frameCount order -> bit-identical to
//                   frame-at-a-time rendering
// and a host-side frame-stream scheduler (submitBatch / pump / retireFront below): consecutive batches share one running
// path pool; the host launches iterations in groups and polls the device's scheduler words (Control).
//
// Path state is structure-of-arrays in float4 groups (16 B per lane per access = 1 KiB per wave
// instruction, fully coalesced).  No MFMA: the path is divergent scalar fp32 + pointer chasing.
#include "../../../include/pt_api.h"
#include "../../../include/pt_debug.h"
#include "pt_device.hpp"

#include <algorithm>
#include <cstddef>
#include <deque>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

using namespace ptd;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(x)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return fail(PT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));    \
    } while (0)

constexpr int BLOCK = 256;
#ifndef SHADE_BLOCK_SIZE
#define SHADE_BLOCK_SIZE 256
#endif
constexpr int SHADE_BLOCK = SHADE_BLOCK_SIZE;     // threads per block of the shading kernel: one scheduler atomic per block per launch
constexpr int TILE_W = 32, TILE_H = 8;
// a BVH whose inner-node records exceed this many bytes (in the 80-B form) makes the hand-written intersect kernel use the 64-B form: what an XCD's 4 MB
// L2 holds beside the triangles and the path state streaming through it (measured: profiles/r04_d_node_record_layout.txt)
#ifndef ASM_NODES_80B_LIMIT
#define ASM_NODES_80B_LIMIT (2 << 20)
#endif

// Path-state accesses stream through the caches once per launch — hundreds of MB per launch through 4 MB of L2 per XCD — while the OTHER stream's intersect
// kernel lives on the BVH's node and triangle lines staying there: every access of a state group (and of the per-frame colour rows) carries the non-temporal
// hint (`nt`: C3 +1.4 %, C4 +0.8 %, C5 +2.3 % with the hit-record stores of pt_extend_gfx950.s hinted too; profiles/r04_v_nontemporal_state.txt)
__device__ __forceinline__ float4 ldS(const float4* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stS(float4* p, float4 v) {
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
}

struct FrameIn { float params[12]; float origin[3]; float rotation[3]; float mouse[3]; };

struct Control {            // device-resident scheduler words shared by the whole batch
    unsigned nextJob;       // next unassigned job
    // exhausted[(j+1)&3] is raised by the shading launch of iteration j when a job pull comes back empty (sticky).  A word is
    // written by iteration j-1 and read by iterations j and j+1 only, so every launch sees stable values:
    //   shading of iteration j writes the dense queue of surviving slots iff exhausted[j&3];
    //   iteration j reads its slots through that queue iff exhausted[(j-1)&3].
    unsigned exhausted[4];
    unsigned jobEnd;        // jobs submitted to the frame stream so far: ids [0, jobEnd) exist (k_submit)
    unsigned needRevive;    // k_submit -> k_revive: dead slots may take jobs again
    unsigned oldestBusy;    // k_scan_inflight: a live slot still works on the oldest unaccumulated batch
    unsigned long long cnt[8];   // PT_CNT_* (device side: segments, nodes, tritests, hitupd, samples, boxtests)
    unsigned qCount[64];    // entries of queue (j&1) at [32*(j&1)]: two words, 128 B apart
    unsigned long long dbg[16];  // developer build (-DPT_PHASE_STATS): trips and active lanes per phase of the intersect kernel
    unsigned busy[8];       // k_scan_inflight: busy[k] = a live slot still works on a frame of the k-th oldest unaccumulated batch
#if defined(PT_PHASE_STATS) || defined(PT_WAVE_STAMPS)
    unsigned long long waveEnd[8192];   // s_memrealtime (100 MHz) at which each wave of the last intersect launch finished
    unsigned long long waveStart[8192]; // ... and started
#endif
};
// pt_extend_gfx950.s reads these two by their byte offsets
static_assert(offsetof(Control, exhausted) == 4 && offsetof(Control, qCount) == 96, "Control layout is part of the hand-written kernel");
__device__ __forceinline__ bool queueIn(const Control* ctl, int iter) { return ctl->exhausted[(iter + 3) & 3] != 0; }

struct State {              // SoA path pool, float4 groups (see header comment)
    float4 *G0, *G1, *G2, *G3, *G4, *G5, *S0, *H;
    unsigned s0Plane;       // STK == 32 (the index stack as ten floats): S0 holds three planes of this many slots — stack slots 0-3, 4-7, 8-9
    uint2* J;               // (frame slot of the stream, accumulator slot) of the slot's job: written when the job starts, read when it ends
    float4* HX;             // (u, v, id) of the closest triangle behind an ellipsoid hit; only for scenes whose ellipsoids carry texture-mapped materials
};

// The frame stream: consecutive batches with the same frame inputs (Parameters, ORIGIN, ROTATION, MOUSE_POS) form ONE job
// sequence, job = streamFrame * nLocal + pixel, so that the path pool never drains between them.  Per-frame data (u_seed,
// the frame's colour row) live in rings of `ringFrames` rows indexed by streamFrame % ringFrames.
struct Batch {
    int W, H, nLocal, nSlots, shardCount;
    unsigned divM, divS;      // job / nLocal == mulhi(job, divM) >> divS for job < 2^31 (nLocal >= 2); divM == 0: nLocal == 1
    unsigned ringFrames;
    const int* seeds;         // device ring, ringFrames
    const int* pixList;       // device, nLocal: global pixel index in tile-major job order
    const unsigned* pixXY;    // device, nLocal: the same pixels as x | y << 16
    float4* colbuf;           // device ring, ringFrames * nSlots
};

// ------------------------------------------------------------------------------------------------ kernels

__global__ void k_frame_setup(DevScene sc, const FrameIn* in, FrameConst* fc, EllipRec* ellip) {
    __shared__ int stk[64];
    if (threadIdx.x != 0) return;
    const float* P = in->params;
    fc->screenSize = P[0]; fc->focalLength = P[1]; fc->resolution = P[2]; fc->screenHratio = P[3]; fc->SAMPLE_RES = P[4];
    fc->MAX_BOUNCES = P[5]; fc->BLUR = P[7]; fc->FOCAL_DISTANCE = P[8]; fc->AUTO_FOCUS = P[11];
    for (int k = 0; k < 3; k++) { fc->origin[k] = in->origin[k]; fc->rotation[k] = in->rotation[k]; fc->mouse[k] = in->mouse[k]; }
    auto rotM = [](float ax, float ay, float az, float* M) {      // rotateX * rotateY * (z != 0 ? rotateZ : I), row-major
        float cx = cos_(ax), sx = sin_(ax), cy = cos_(ay), sy = sin_(ay);
        float RX[9] = {1, 0, 0, 0, cx, sx, 0, -sx, cx};
        float RY[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
        float RZ[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (az != 0.0f) { float cz = cos_(az), sz = sin_(az); RZ[0] = cz; RZ[1] = sz; RZ[3] = -sz; RZ[4] = cz; }
        float T[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[3 * i + j] = RX[3 * i] * RY[j] + RX[3 * i + 1] * RY[3 + j] + RX[3 * i + 2] * RY[6 + j];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[3 * i + j] = T[3 * i] * RZ[j] + T[3 * i + 1] * RZ[3 + j] + T[3 * i + 2] * RZ[6 + j];
    };
    rotM(in->rotation[0], in->rotation[1], in->rotation[2], fc->camRot);
    for (int i = 0; i < sc.numEllip; i++) {
        EllipRec& E = ellip[i];
        float rx = E.rot[0], ry = E.rot[1], rz = E.rot[2];
        E.rotated = length(v3(rx, ry, rz)) > 0.0f ? 1 : 0;       // frag.glsl:613
        if (E.rotated) {
            float cx = cos_(rx), sx = sin_(rx), cy = cos_(ry), sy = sin_(ry), cz = cos_(rz), sz = sin_(rz);
            float c0[3] = {cy * cz, cx * sz + cz * sx * sy, sx * sz - cx * cz * sy};      // rotateBack columns, :291-295
            float c1[3] = {-cy * sz, cx * cz - sx * sy * sz, cz * sx + cx * sy * sz};
            float c2[3] = {sy, -cy * sx, cx * cy};
            for (int r = 0; r < 3; r++) { E.RB[3 * r] = c0[r]; E.RB[3 * r + 1] = c1[r]; E.RB[3 * r + 2] = c2[r]; }
            rotM(rx, ry, rz, E.R);
        }
    }
    __threadfence();
    float mid = -1.0f;
    if (fc->AUTO_FOCUS == 1.0f) {                                  // :901-906
        DevScene s2 = sc; s2.ldsNodes = 0; s2.ldsTris = 0; s2.ellip = ellip;
        vec3 fwd = vecmat(v3(0.0f, 0.0f, 1.0f), fc->camRot);
        float t, u, v; int prim; Counters c;
        intersectScene<false>(s2, v3(in->origin[0], in->origin[1], in->origin[2]), fwd, stk, 1, nullptr, nullptr, t, u, v, prim, c);
        mid = (t < 1e25f) ? t : -1.0f;                             // result.distance (:641,:651)
    }
    fc->midToScene = mid;
    fc->focus = (fc->AUTO_FOCUS == 1.0f && mid > 0.0f) ? mid : fc->FOCAL_DISTANCE;
}

// Path state in memory (16 B per lane per group, consecutive lanes on consecutive slots):
//   G0 = O.xyz, D.x    G1 = D.yz, rngState, flags    G2 = throughput.rgb, index-stack codes (3-bit scenes)    G4 = sum over samples.rgb, pixel x | y << 16
//   G3 = incLight.rgb of the running sample, touched only while FL_INCNZ (pt_device.hpp)      J = (frame slot, accumulator slot), job start / job end only
//   G5 = RAY_ENTER_LOCATION, DISTANCE_TRAVELED and S0 = index-stack codes of an 8-bit scene: scenes with transmissive materials only
__device__ __forceinline__ void storeStack32(const State& st, unsigned i, const Path& p) {
    stS(st.S0 + i, make_float4(__uint_as_float(p.sk[0]), __uint_as_float(p.sk[1]), __uint_as_float(p.sk[2]), __uint_as_float(p.sk[3])));
    stS(st.S0 + st.s0Plane + i, make_float4(__uint_as_float(p.sk[4]), __uint_as_float(p.sk[5]), __uint_as_float(p.sk[6]), __uint_as_float(p.sk[7])));
    stS(st.S0 + 2 * (size_t)st.s0Plane + i, make_float4(__uint_as_float(p.sk[8]), __uint_as_float(p.sk[9]), 0.0f, 0.0f));
}
template <int STK>
__device__ __forceinline__ void storePath(const State& st, unsigned i, const Path& p) {
    stS(st.G0 + i, make_float4(p.O.x, p.O.y, p.O.z, p.D.x));
    stS(st.G1 + i, make_float4(p.D.y, p.D.z, __uint_as_float(p.rng), __uint_as_float(packFlags(p))));
    stS(st.G2 + i, make_float4(p.col.x, p.col.y, p.col.z, __uint_as_float(p.sc0)));
    stS(st.G3 + i, make_float4(p.inc.x, p.inc.y, p.inc.z, 0.0f));      // (job starts only; directDiffuse reads the group for every lane)
    stS(st.G4 + i, make_float4(p.sum.x, p.sum.y, p.sum.z, __uint_as_float(p.pix)));
    st.J[i] = make_uint2(p.fi, p.ls);
    if (STK) st.G5[i] = make_float4(p.enter.x, p.enter.y, p.enter.z, p.dist);
    if (STK == 8) st.S0[i] = make_float4(__uint_as_float(p.sc0), __uint_as_float(p.sc1), __uint_as_float(p.sc2), 0.0f);
    if (STK == 32) storeStack32(st, i, p);
}

// A new pixel-frame job (one fragment-shader invocation): fresh "globals" (SURVEY.md Q-1), rngState = index + u_seed.
// The caller follows up with startSample (kept separate so that k_shade has a single startSample site).
__device__ __forceinline__ void startJob(const Batch& b, const FrameConst& fc, unsigned job, Path& p) {
    unsigned fi = b.divM ? (__umulhi(job, b.divM) >> b.divS) : job;      // job / nLocal without an integer division
    unsigned k = job - fi * (unsigned)b.nLocal;
    unsigned xy = b.pixXY[k];
    int px = (int)(xy & 0xffffu), py = (int)(xy >> 16);
    uint32_t index;
    pixelIndex(fc, b.W, b.H, px, py, index);
    p.pix = xy; p.fi = fi;
    p.ls = (b.shardCount == 1) ? (unsigned)(py * b.W + px) : k;
    p.rng = index + (uint32_t)b.seeds[fi % b.ringFrames];
    p.sum = v3(0.0f);
    p.sample = 0;
    p.applyAbs = false; p.inObj = false;
    p.enter = v3(0.0f); p.dist = 0.0f;
    p.sc0 = 0u; p.sc1 = 0u; p.sc2 = 0u;                         // every slot of the index stack 0.0 (dictionary code 0)
#pragma unroll
    for (int k = 0; k < 10; k++) p.sk[k] = 0u;
    p.stackSize = 0;
    p.incNZ = false; p.inc = v3(0.0f);
    p.alive = true;
}

// Every dead slot of the pool asks for a job (one scheduler atomic per block) and, if one is left, starts it: the start of
// a frame stream (all slots dead) and the restart after the pool ran dry between two batches.
template <int STK, bool FAST>
__global__ void __launch_bounds__(BLOCK) k_revive(Batch b, const FrameConst* fcp, State st, int nSlots, Control* ctl) {
    __shared__ unsigned sCnt[BLOCK / 64], sBase;
    const unsigned mode = ctl->needRevive;
    if (!mode) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned i = blockIdx.x * BLOCK + threadIdx.x;
    if (mode == 2u) {                                              // start of a stream: every slot is dead, slot i takes job i
        if (i >= (unsigned)nSlots || i >= ctl->jobEnd) return;
        const FrameConst& fc = *fcp;
        Path p;
        startJob(b, fc, i, p);
        startSample<STK, FAST>(fc, b.W, b.H, (int)(p.pix & 0xffffu), (int)(p.pix >> 16), p);
        storePath<STK>(st, i, p);
        st.H[i] = make_float4(1e30f, 0.0f, 0.0f, __int_as_float(PRIM_NONE));
        return;
    }
    const bool want = i < (unsigned)nSlots && !(__float_as_uint(st.G1[i].w) & FL_ALIVE);
    unsigned long long mask = __ballot(want);
    if (lane == 0) sCnt[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) total += sCnt[w];
        sBase = total ? atomicAdd(&ctl->nextJob, total) : 0u;
    }
    __syncthreads();
    if (!want) return;
    unsigned off = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) off += (w < wave) ? sCnt[w] : 0u;
    const unsigned job = sBase + off + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    if (job >= ctl->jobEnd) return;                               // stays dead
    const FrameConst& fc = *fcp;
    Path p;
    startJob(b, fc, job, p);
    startSample<STK, FAST>(fc, b.W, b.H, (int)(p.pix & 0xffffu), (int)(p.pix >> 16), p);
    storePath<STK>(st, i, p);
    st.H[i] = make_float4(1e30f, 0.0f, 0.0f, __int_as_float(PRIM_NONE));
}

// rayScene for every live path slot.  Dynamic LDS: [nodes 4*ldsNodes float4][tris 3*ldsTris float4][stack depth*BLOCK int]
template <bool COUNT>
__global__ void __launch_bounds__(BLOCK) k_extend(DevScene sc, State st, const unsigned* qIn, int iter, int nSlots, Control* ctl) {
    extern __shared__ float4 smem[];
    float4* ldsN = smem;
    float4* ldsT = smem + 4 * sc.ldsNodes;
    int* stkBase = reinterpret_cast<int*>(smem + 4 * sc.ldsNodes + 3 * sc.ldsTris);
    const unsigned* queue = queueIn(ctl, iter) ? qIn : nullptr;
    unsigned n = queue ? ctl->qCount[32 * (iter & 1)] : (unsigned)nSlots;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->qCount[32 * ((iter + 1) & 1)] = 0;     // cursor of the queue this iteration's shading may write
    if (blockIdx.x * BLOCK >= n) return;                       // whole block beyond the live range
    for (int k = threadIdx.x; k < 4 * sc.ldsNodes; k += BLOCK) ldsN[k] = sc.nodes[k];
    for (int k = threadIdx.x; k < 3 * sc.ldsTris; k += BLOCK) ldsT[k] = sc.tris[k];
    __syncthreads();
    unsigned q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n) return;
    unsigned i = queue ? queue[q] : q;
    float4 g0 = st.G0[i], g1 = st.G1[i];
    if (!(__float_as_uint(g1.w) & FL_ALIVE)) return;
    float t, u, v; int prim; Counters c;
    const unsigned fl = __float_as_uint(g1.w);
    intersectScene<COUNT>(sc, v3(g0.x, g0.y, g0.z), v3(g0.w, g1.x, g1.y), stkBase + threadIdx.x, BLOCK, ldsN, ldsT, t, u, v, prim, c,
                          (fl & FL_PROBE) != 0, probeObjOf(fl), st.HX ? st.HX + i : nullptr);
    st.H[i] = make_float4(t, u, v, __int_as_float(prim));
    if (COUNT) {
        atomicAdd(&ctl->cnt[PT_CNT_NODES], (unsigned long long)c.nodes);
        atomicAdd(&ctl->cnt[PT_CNT_TRITESTS], (unsigned long long)c.tritests);
        atomicAdd(&ctl->cnt[PT_CNT_HITUPD], (unsigned long long)c.hitupd);
        atomicAdd(&ctl->cnt[PT_CNT_BOXTESTS], (unsigned long long)c.boxtests);
    }
}

// Persistent, phase-selecting form of k_extend (the production intersect kernel).
//  * grid sized to the machine; every block stages the top of the BVH (and, when they fit, the triangles) into LDS
//    ONCE and then streams its share of the rays through it;
//  * per wave: a lane whose ray is finished takes the next ray of the wave's static range (ballot + prefix
//    popcount, no atomics) instead of idling until the slowest lane is done;
//  * every trip through the loop the wave issues ONE instruction stream — the inner-node step or the leaf step —
//    whichever more lanes are waiting for (ballot + popcount); the other lanes keep their entry for a later trip.
//    The kernel is VALU-issue bound (profiles/): what counts is lanes busy per issued instruction;
//  * the entry the reference would push last and pop next (the near child) stays in a register; only the far child
//    goes through the LDS stack.
// Per ray the sequence of visited nodes / tested triangles and every comparison are exactly rayBVH's (frag.glsl:
// 452-537): a lane never steps past a pending leaf, so `closest` evolves in the reference's order.
// `cur` = the entry this lane will process next (the virtual top of rayBVH's stack):
//   >= 0 and < CUR_NONE  inner node   |   < 0  leaf: first triangle record is -(cur+1)
//   CUR_NONE  ray alive, BVH of the current object exhausted   |   CUR_IDLE  lane has no ray
// (activity is folded into `cur` so that every wave-level vote is ONE v_cmp writing an SGPR pair)
//   CUR_DONE  ray finished, its hit record still in the lane's registers (stored at the wave's next refill)
// lanes set in a wave mask, as a 32-bit SCALAR: left to __popcll the compiler keeps the count 64 bits wide and compares two counts with
// v_cmp_*_u64 (there is no 64-bit scalar ordered compare) — six vector instructions per outer trip of the intersect kernel
__device__ __forceinline__ int wavePop(unsigned long long m) { int n; asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc"); return n; }
constexpr int CUR_NONE = 0x7ffffffd;
constexpr int CUR_DONE = 0x7ffffffe;
constexpr int CUR_IDLE = 0x7fffffff;
// Traversal-stack entries (pending far children) live in LDS, one column per lane.  Their width decides how many blocks a CU holds:
//   short     trees below 32767 inner nodes / triangle records
//   Packed18  up to 131071: the low 16 bits in LDS, bits 16-17 in a per-lane shift register (two VGPRs, 2 bits per level, 32 levels)
//   int       anything larger
struct Packed18 {};
template <typename T> struct StackElem { typedef T type; };
template <> struct StackElem<Packed18> { typedef unsigned short type; };

#ifndef PT_EP_WAVES
#define PT_EP_WAVES 8        // waves per SIMD the register allocation must leave room for (8 blocks of 256 threads per CU)
#endif
template <bool COUNT, typename StackT, int TPB, bool RARE>
__global__ void __launch_bounds__(TPB, PT_EP_WAVES) k_extend_persist(DevScene sc, State st, const unsigned* qIn, int iter, int nSlots,
                                                       Control* ctl, int refillMin, int keepEighths, int nObjLds, int noneMin) {
    extern __shared__ float4 smem[];
    typedef typename StackElem<StackT>::type ElemT;
    constexpr bool PACKED = __is_same(StackT, Packed18);
    unsigned hiA = 0u, hiB = 0u;                                   // Packed18: bits 16-17 of the stacked entries, newest in hiA[1:0]
    float4* ldsN = smem;
    float4* ldsT = smem + 4 * sc.ldsNodes;
    // per-lane root-box distances (rayNode(o,d,root) of :468 depends on the ray only, so it is evaluated once per ray
    // when the lane takes the ray — all refilled lanes together — and only compared against `closest` later)
    float* rootDist = reinterpret_cast<float*>(smem + 4 * sc.ldsNodes + 3 * sc.ldsTris) + threadIdx.x;
    // the first nObjLds object roots (box + reference, 32 B each) are read by every refill and every next-object step: LDS copies
    float4* rootsL = reinterpret_cast<float4*>(reinterpret_cast<float*>(smem + 4 * sc.ldsNodes + 3 * sc.ldsTris) + nObjLds * TPB);
    // the slot a lane's ray came from is needed again only when the ray retires: parked in LDS, not in a register
    unsigned* slotL = reinterpret_cast<unsigned*>(rootsL + 2 * nObjLds) + threadIdx.x;
    ElemT* stk = reinterpret_cast<ElemT*>(reinterpret_cast<unsigned*>(rootsL + 2 * nObjLds) + TPB) + threadIdx.x;
    for (int k = threadIdx.x; k < 2 * nObjLds; k += TPB) rootsL[k] = reinterpret_cast<const float4*>(sc.roots)[k];
    for (int k = threadIdx.x; k < 4 * sc.ldsNodes; k += TPB) ldsN[k] = sc.nodes[k];
    for (int k = threadIdx.x; k < 3 * sc.ldsTris; k += TPB) ldsT[k] = sc.tris[k];
    __syncthreads();
    const unsigned* queue = queueIn(ctl, iter) ? qIn : nullptr;
    const unsigned n = queue ? ctl->qCount[32 * (iter & 1)] : (unsigned)nSlots;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->qCount[32 * ((iter + 1) & 1)] = 0;     // cursor of the queue this iteration's shading may write
    [[maybe_unused]] const int lane = threadIdx.x & 63;
    const unsigned nWaves = gridDim.x * (TPB / 64), waveId = __builtin_amdgcn_readfirstlane(blockIdx.x * (TPB / 64) + (threadIdx.x >> 6));   // scalar: pos/end live in SGPRs
    const unsigned per = (((n + nWaves - 1) / nWaves) + 63u) & ~63u;       // static range of this wave, 64-aligned -> coalesced first fill
    unsigned pos = waveId * per;
    const unsigned end = min(pos + per, n);
    vec3 o = v3(0.0f), d = v3(0.0f), invD = v3(0.0f);
    float closest = 1e30f, hu = 0.0f, hv = 0.0f;
    int prim = PRIM_NONE, ob = 0, obEnd = 0, sp = 0, cur = CUR_IDLE;
    bool probe = false;
    unsigned slot = 0;
    Counters c;
#if defined(PT_PHASE_STATS) || defined(PT_WAVE_STAMPS)
    const unsigned long long tStart = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PT_PHASE_STATS
    unsigned long long ps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // {trips, active lanes} x {refill, next-object/retire, inner, leaf}, outer trips, live lanes at outer trips
#define PS(k, lanes) do { ps[2 * (k)]++; ps[2 * (k) + 1] += (unsigned long long)(lanes); } while (0)
    unsigned long long pt_[5] = {0, 0, 0, 0, 0}, tPrev = __builtin_amdgcn_s_memtime();      // shader-clock cycles per phase: refill, next-object, inner, leaf, votes and the rest
#define PTIME(k) do { const unsigned long long tNow_ = __builtin_amdgcn_s_memtime(); pt_[k] += tNow_ - tPrev; tPrev = tNow_; } while (0)
#else
#define PS(k, lanes) do { } while (0)
#define PTIME(k) do { } while (0)
#endif
    for (;;) {
        // ---- refill idle lanes from the wave's range
        unsigned long long idle = __ballot(cur >= CUR_DONE);                // lanes without a ray in flight
        int nIdle = wavePop(idle);
        PS(4, 64 - nIdle);
        if (pos < end && nIdle >= refillMin) {                              // wave-uniform
            PS(0, min(nIdle, (int)(end - pos)));
            PTIME(4);
            if (cur >= CUR_DONE) {
                // the hit records of the rays that retired since the last refill leave together: one store instruction for all of them,
                // neighbouring slots of one refill generation close in time (the lanes retire out of order)
                if (cur == CUR_DONE) { st.H[*slotL] = make_float4(closest, hu, hv, __int_as_float(prim)); cur = CUR_IDLE; }
                // rank of this lane among the idle ones: v_mbcnt (set bits of the mask below the lane), no lane-mask registers
                unsigned q = pos + __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                if (q < end) {
                    slot = queue ? queue[q] : q;
                    *slotL = slot;
                    float4 g0 = st.G0[slot], g1 = st.G1[slot];
                    // both groups in ONE round trip (left alone the compiler fetches the flags word first and the rest behind the branch)
                    asm volatile("" : "+v"(g0.x), "+v"(g0.y), "+v"(g0.z), "+v"(g0.w), "+v"(g1.x), "+v"(g1.y), "+v"(g1.z), "+v"(g1.w));
                    const unsigned fl = __float_as_uint(g1.w);
                    if (fl & FL_ALIVE) {
                        d = v3(g0.w, g1.x, g1.y);
                        probe = RARE && (fl & FL_PROBE) != 0;             // directDiffuse's thickness probe: rayBVH called directly (:668); only RAYTRACING == 0 makes them
                        o = probe ? v3(g0.x, g0.y, g0.z) : madd(d, 1e-4f, v3(g0.x, g0.y, g0.z));   // o = o + 1e-4*d  (:549)
                        invD = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        ob = probe ? probeObjOf(fl) : 0;
                        obEnd = probe ? ob + 1 : sc.numObj;
                        closest = 1e30f; hu = 0.0f; hv = 0.0f; prim = PRIM_NONE; sp = 0; cur = CUR_NONE;
                        for (int k = 0; k < nObjLds; k += 2) {              // two root boxes per packed-f32 test, like the two children of a node
                            const int kb = min(k + 1, nObjLds - 1);
                            const float4 a0 = rootsL[2 * k], a1 = rootsL[2 * k + 1], b0 = rootsL[2 * kb], b1 = rootsL[2 * kb + 1];
                            ObjRoot A, B;
                            A.bmin[0] = a0.x; A.bmin[1] = a0.y; A.bmin[2] = a0.z; A.bmax[0] = a0.w; A.bmax[1] = a1.x; A.bmax[2] = a1.y;
                            B.bmin[0] = b0.x; B.bmin[1] = b0.y; B.bmin[2] = b0.z; B.bmax[0] = b0.w; B.bmax[1] = b1.x; B.bmax[2] = b1.y;
                            float da, db;
                            rayBox2(o, invD, make_float4(A.bmin[0], B.bmin[0], A.bmin[1], B.bmin[1]), make_float4(A.bmin[2], B.bmin[2], A.bmax[0], B.bmax[0]),
                                    make_float4(A.bmax[1], B.bmax[1], A.bmax[2], B.bmax[2]), da, db);
                            rootDist[k * TPB] = da;
                            if (k + 1 < nObjLds) rootDist[(k + 1) * TPB] = db;
                        }
                    }
                }
            }
            pos += (unsigned)nIdle;
            nIdle = wavePop(__ballot(cur >= CUR_DONE));
            PTIME(0);
        }
        if (nIdle == 64) { if (pos >= end) break; continue; }
        // ---- lanes whose BVH is exhausted: next object (root box test :468), else ellipsoids + retire.  Like the two traversal
        // phases below this one is worth a trip only for enough lanes: it runs when noneMin lanes wait for it, or when it is the
        // most wanted of the three
        int nInner = wavePop(__ballot((unsigned)cur < (unsigned)CUR_NONE));   // cur in [0, CUR_NONE)
        int nLeaf = wavePop(__ballot(cur < 0));
        const int nNone = wavePop(__ballot(cur == CUR_NONE));
        if (nNone >= noneMin || (nNone > 0 && nNone >= nInner && nNone >= nLeaf)) {
            PS(1, nNone);
            PTIME(4);
            if (cur == CUR_NONE) {
                while (ob < obEnd) {
                    float rd; int rref;
                    if (ob < nObjLds) { rd = rootDist[ob * TPB]; rref = __float_as_int(rootsL[2 * ob + 1].z); }
                    else { const ObjRoot R = sc.roots[ob]; rd = rayBox(o, invD, R.bmin[0], R.bmin[1], R.bmin[2], R.bmax[0], R.bmax[1], R.bmax[2]); rref = R.ref; }
                    ob++;
                    if (COUNT) c.boxtests++;
                    if (rd > closest) continue;
                    if (rref == REF_EMPTY) { if (COUNT) c.nodes++; continue; }
                    cur = rref;
                    break;
                }
                if (cur == CUR_NONE) {                                      // all BVHs done: ellipsoids (:606-631), then retire the ray
                    for (int i = 0; i < (probe ? 0 : sc.numEllip); i++) {
                        const EllipRec& E = sc.ellip[i];
                        vec3 cc = v3(E.c[0], E.c[1], E.c[2]);
                        float t;
                        if (E.rotated) t = rayEllipsoid(vecmat(o, E.R), vecmat(d, E.R), cc, E.r, E.st[0], E.st[1], E.st[2]);
                        else t = rayEllipsoid(o, d, cc, E.r, E.st[0], E.st[1], E.st[2]);
                        if (t < closest) {
                            if (!(prim & PRIM_ELLIPSOID) || prim == PRIM_NONE) {                            // see intersectScene
                                if (RARE && st.HX) st.HX[*slotL] = make_float4(hu, hv, __int_as_float(prim), 0.0f);
                                hu = __int_as_float(prim);
                            }
                            closest = t; prim = PRIM_ELLIPSOID | i;
                        }
                    }
                    cur = CUR_DONE;
                }
            }
            nInner = wavePop(__ballot((unsigned)cur < (unsigned)CUR_NONE));
            nLeaf = wavePop(__ballot(cur < 0));
            PTIME(1);
        }
        PTIME(4);
        if (nInner >= nLeaf && nInner > 0) {
            // ---- inner-node steps (:521-532); repeated while most of the lanes that started the phase still sit on inner nodes
            const int keepGoing = (nInner * keepEighths) >> 3;
            do {
                PS(2, nInner);
                if ((unsigned)cur < (unsigned)CUR_NONE) {
                    float4 q0, q1, q2, q3;
                    loadNode2(sc, ldsN, cur, q0, q1, q2, q3);
                    if (COUNT) { c.nodes++; c.boxtests += 2; }
                    float Ld, Rd;
                    rayBox2(o, invD, q0, q1, q2, Ld, Rd);
                    const int lref = __float_as_int(q3.x), rref = __float_as_int(q3.y);
                    if (COUNT) { if (Ld < closest && lref == REF_EMPTY) c.nodes++; if (Rd < closest && rref == REF_EMPTY) c.nodes++; }
                    // :525-531 pushes the farther child first, so the nearer one (ties: the left one) is popped next
                    const bool rNear = Ld > Rd;
                    const int nearRef = rNear ? rref : lref, farRef = rNear ? lref : rref;
                    const float nearD = rNear ? Rd : Ld, farD = rNear ? Ld : Rd;
                    const bool nearOk = nearD < closest && nearRef != REF_EMPTY, farOk = farD < closest && farRef != REF_EMPTY;
                    if (nearOk) {
                        cur = nearRef;
                        if (farOk) {
                            stk[sp * TPB] = (ElemT)farRef; sp++;
                            if (PACKED) { hiB = __builtin_amdgcn_alignbit(hiB, hiA, 30); hiA = (hiA << 2) | (((unsigned)farRef >> 16) & 3u); }
                        }
                    } else if (farOk) {
                        cur = farRef;
                    } else if (sp > 0) {
                        cur = (int)stk[(--sp) * TPB];
                        if (PACKED) { cur = (int)(((unsigned)cur | (hiA << 16)) << 14) >> 14; hiA = __builtin_amdgcn_alignbit(hiB, hiA, 2); hiB >>= 2; }
                    } else {
                        cur = CUR_NONE;
                    }
                }
                nInner = wavePop(__ballot((unsigned)cur < (unsigned)CUR_NONE));
            } while (nInner > keepGoing);
            PTIME(2);
        } else if (nLeaf > 0) {
            // ---- leaf steps: one triangle of the pending leaf per step (:483-520); repeated while most lanes still have
            // triangles left in their leaf (the reference's builder can leave many triangles in one leaf)
            const int keepGoing = (nLeaf * keepEighths) >> 3;
            int nMore;
            do {
                PS(3, __popcll(__ballot(cur < 0)));
                bool more = false;
                if (cur < 0) {
                    int ti = -(cur + 1);
                    float4 t0, t1, t2;
                    loadTri(sc, ldsT, ti, t0, t1, t2);
                    unsigned idl = __float_as_uint(t2.y);
                    float t, u, v;
                    if (COUNT) c.tritests++;
                    rayTri(o, d, v3(t0.x, t0.y, t0.z), v3(t0.w, t1.x, t1.y), v3(t1.z, t1.w, t2.x), t, u, v);
                    if (t > 0.0f && t < closest) {                          // :489
                        closest = t; hu = u; hv = v; prim = (int)(idl & 0x7fffffffu);
                        if (COUNT) c.hitupd++;
                    }
                    if (idl >> 31) {                                        // last triangle of the leaf: this node is done
                        if (COUNT) c.nodes++;
                        if (sp > 0) {
                            cur = (int)stk[(--sp) * TPB];
                            if (PACKED) { cur = (int)(((unsigned)cur | (hiA << 16)) << 14) >> 14; hiA = __builtin_amdgcn_alignbit(hiB, hiA, 2); hiB >>= 2; }
                        } else cur = CUR_NONE;
                    } else {
                        cur = cur - 1;                                      // next triangle record of the same leaf
                        more = true;
                    }
                }
                nMore = wavePop(__ballot(more));
            } while (nMore > keepGoing);
            PTIME(3);
        }
    }
    if (cur == CUR_DONE) st.H[*slotL] = make_float4(closest, hu, hv, __int_as_float(prim));      // the rays that retired after the last refill
    if (COUNT) {
        atomicAdd(&ctl->cnt[PT_CNT_NODES], (unsigned long long)c.nodes);
        atomicAdd(&ctl->cnt[PT_CNT_TRITESTS], (unsigned long long)c.tritests);
        atomicAdd(&ctl->cnt[PT_CNT_HITUPD], (unsigned long long)c.hitupd);
        atomicAdd(&ctl->cnt[PT_CNT_BOXTESTS], (unsigned long long)c.boxtests);
    }
#ifdef PT_PHASE_STATS
    if (lane == 0) { for (int k = 0; k < 10; k++) atomicAdd(&ctl->dbg[k], ps[k]); for (int k = 0; k < 5; k++) atomicAdd(&ctl->dbg[10 + k], pt_[k]); }
#endif
#if defined(PT_PHASE_STATS) || defined(PT_WAVE_STAMPS)
    if (lane == 0 && waveId < 8192) { ctl->waveEnd[waveId] = __builtin_amdgcn_s_memrealtime(); ctl->waveStart[waveId] = tStart; }
#endif
#undef PS
#undef PTIME
}

// trace() loop body + sample/job bookkeeping for every live path slot.
// Thread t of a block handles queue position (or slot) blockIdx*256 + t: every access to the path state is 16 B per lane at
// consecutive addresses.  (An earlier form sorted the block's slots by outcome — hit / miss / dead — through LDS so that waves
// ran one instruction stream each; once the kernel had become memory-bound the permuted accesses cost more than the
// divergence saved: 2-4 % per step on C2-C5, profiles/; likewise the dense LDS-listed pass that used to compute the camera
// rays of new samples for the whole block.)
template <int STK, bool STATS, bool DIRECT, bool TEX, bool FAST = false>
// (the texture-map variants are asked for 5 waves per SIMD: left alone the compiler spends 104-110 registers on them, 4 waves; at 96 it spills two and a
//  scene with a map on every material gains 3 %: profiles/r05_k_textured_materials.txt.  The float-stack variant would spill ten: left alone.)
__global__ void __launch_bounds__(SHADE_BLOCK, (TEX && !STATS && STK != 32) ? 5 : 1) k_shade(DevScene sc, Batch b, const FrameConst* fcp, State st, const unsigned* qIn, unsigned* qOut, int iter,
                                                 int nSlots, Control* ctl) {
    constexpr bool TRANS = STK != 0;
    __shared__ unsigned sCntA[SHADE_BLOCK / 64], sCntB[SHADE_BLOCK / 64], sBase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    const FrameConst& fc = *fcp;
    const unsigned* queue = queueIn(ctl, iter) ? qIn : nullptr;
    const unsigned n = queue ? ctl->qCount[32 * (iter & 1)] : (unsigned)nSlots;
    const bool writeQueue = ctl->exhausted[iter & 3] != 0;        // job supply ran dry before this iteration: pack the survivors
    const unsigned jobEnd = ctl->jobEnd;
    if (writeQueue && blockIdx.x == 0 && threadIdx.x == 0) ctl->exhausted[(iter + 1) & 3] = 1u;     // sticky, also through an empty launch
    if (blockIdx.x * SHADE_BLOCK >= n) return;                           // the grid is sized by the host's last known bound
    // ---- 1. the slot and its state: every load of the segment is issued here, in one batch — the kernel's critical path is
    // memory round trips, not bytes.  Only the groups few lanes need wait for the flags: incLight (G3) and the medium-entry group (G5).
    const unsigned q = blockIdx.x * SHADE_BLOCK + threadIdx.x;
    const bool valid = q < n;
    const unsigned i = valid ? (queue ? queue[q] : q) : 0u;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0, g3 = g0, g4 = g0, h = g0, s0 = g0, s1 = g0, s2 = g0, g5 = g0;
    // Path tracing decides BEFORE it shades whether the segment ends its sample (segmentEndsSample: the hit record, the throughput and the bounce count say so):
    // the sum over the job's samples (G4, 16 B) and the job words (J) are then fetched by the lanes whose sample / job ends only — a segment in four — and the job
    // pull runs while those loads and the shading records are on their way.  directDiffuse learns it from the material (a subsurface hit goes on as a probe): old order.
    constexpr bool EARLY = !DIRECT;
    if (valid) {
        g1 = ldS(st.G1 + i); g0 = ldS(st.G0 + i); h = ldS(st.H + i); g2 = ldS(st.G2 + i);
        if (!EARLY) g4 = ldS(st.G4 + i);
        if (STK == 8 || STK == 32) s0 = ldS(st.S0 + i);
        if (STK == 32) { s1 = ldS(st.S0 + st.s0Plane + i); s2 = ldS(st.S0 + 2 * (size_t)st.s0Plane + i); }
        if (DIRECT) g3 = ldS(st.G3 + i);
    }
    const bool live = valid && (__float_as_uint(g1.w) & FL_ALIVE);
    Path p;
    p.alive = false;
    bool jobDone = false, needStart = false;
    unsigned nSamp = 0;
    // Groups are written back only when this segment changed them: sum (G4) at sample end; RAY_ENTER_LOCATION / DISTANCE_TRAVELED (G5) when the
    // transmission lobe won; incLight (G3) when it is not zero and not what memory holds.
    bool sampleDone = false, newJob = false, isProbe = false, incInMemory = false;
    float4 g3in = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint2 jobWords = make_uint2(0u, 0u);
    if (live) {
        unpackFlags(p, __float_as_uint(g1.w));
        // incLight of the running sample: +0.0 in every component unless the path has met an emitter and gone on (FL_INCNZ): only those lanes fetch it.
        // (directDiffuse parks its probe state in the group: that mode reads and writes it for every lane.)
        incInMemory = DIRECT || p.incNZ;
        if (!DIRECT && p.incNZ) g3 = ldS(st.G3 + i);
        if (TRANS) {
            // RAY_ENTER_LOCATION / DISTANCE_TRAVELED are read only while the path is inside a medium or owes an absorption term;
            // a transmission event on any other path fetches them late (shadeSegment), which is rare
            p.g5loaded = p.inObj || p.applyAbs;
            if (p.g5loaded) g5 = ldS(st.G5 + i);
        }
        p.O = v3(g0.x, g0.y, g0.z); p.D = v3(g0.w, g1.x, g1.y); p.rng = __float_as_uint(g1.z);
        p.col = v3(g2.x, g2.y, g2.z);
        if (EARLY) {
            sampleDone = segmentEndsSample<FAST>(fc, p, h.x, __float_as_int(h.w));
            if (sampleDone) {
                g4 = ldS(st.G4 + i);
                nSamp = 1;
                if ((float)(p.sample + 1) < fc.SAMPLE_RES) needStart = true;      // loop condition :898
                else { jobDone = true; jobWords = st.J[i]; }
            }
        }
        p.inc = incInMemory ? v3(g3.x, g3.y, g3.z) : v3(0.0f);
        p.sum = v3(g4.x, g4.y, g4.z); p.pix = __float_as_uint(g4.w);      // (path tracing: of the lanes whose sample ends; the others neither read nor write them)
        p.fi = 0u; p.ls = 0u;                                       // the job's (frame slot, accumulator slot) wait in J until the job ends
        p.enter = v3(g5.x, g5.y, g5.z); p.dist = g5.w; p.g5dirty = false;
        if (!TRANS) p.g5loaded = true;
        g3in = make_float4(p.inc.x, p.inc.y, p.inc.z, 0.0f);
        if (STK == 8) { p.sc0 = __float_as_uint(s0.x); p.sc1 = __float_as_uint(s0.y); p.sc2 = __float_as_uint(s0.z); }
        else { p.sc0 = __float_as_uint(g2.w); p.sc1 = 0u; p.sc2 = 0u; }
        if (STK == 32) {
            p.sk[0] = __float_as_uint(s0.x); p.sk[1] = __float_as_uint(s0.y); p.sk[2] = __float_as_uint(s0.z); p.sk[3] = __float_as_uint(s0.w);
            p.sk[4] = __float_as_uint(s1.x); p.sk[5] = __float_as_uint(s1.y); p.sk[6] = __float_as_uint(s1.z); p.sk[7] = __float_as_uint(s1.w);
            p.sk[8] = __float_as_uint(s2.x); p.sk[9] = __float_as_uint(s2.y);
        }
        isProbe = DIRECT && p.probe;
        if (DIRECT) {
            sampleDone = directSegment<TEX>(sc, p, h.x, h.y, h.z, __float_as_int(h.w), st.HX, i);      // RAYTRACING == 0 (frag.glsl:911-912)
            if (sampleDone) {
                p.sum = p.sum + p.inc;                                 // col += trace(...)  (:910)
                p.sample++;
                nSamp = 1;
                if ((float)p.sample < fc.SAMPLE_RES) needStart = true;      // loop condition :898
                else { jobDone = true; jobWords = st.J[i]; }
            }
        }
    }
    // ---- 2. job pull: ballot + prefix count per wave, aggregated over the block's 4 waves through LDS, so the
    // scheduler word sees ONE atomic per 256 lanes per launch (a single address sustains only ~90 atomics/us)
    unsigned long long mask = __ballot(jobDone);
    if (lane == 0) sCntA[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < SHADE_BLOCK / 64; w++) total += sCntA[w];
        sBase = total ? (writeQueue ? jobEnd : atomicAdd(&ctl->nextJob, total)) : 0u;       // known dry: no need to ask
        if (total && sBase + total > jobEnd) ctl->exhausted[(iter + 1) & 3] = 1u;                      // an empty pull: the tail begins
    }
    __syncthreads();
    unsigned nextJob = 0u;
    bool gotJob = false;
    if (jobDone) {
        unsigned off = 0;
#pragma unroll
        for (int w = 0; w < SHADE_BLOCK / 64; w++) off += (w < wave) ? sCntA[w] : 0u;
        nextJob = sBase + off + (unsigned)__popcll(mask & ltMask);
        gotJob = nextJob < jobEnd;
    }
    // ---- 2b. tail of the batch: the slots that stay alive go into the next iteration's dense queue (one atomic per block)
    if (writeQueue) {
        const bool keep = live && p.alive && !(jobDone && !gotJob);
        unsigned long long mk = __ballot(keep);
        if (lane == 0) sCntB[wave] = (unsigned)__popcll(mk);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
#pragma unroll
            for (int w = 0; w < SHADE_BLOCK / 64; w++) total += sCntB[w];
            sBase = total ? atomicAdd(&ctl->qCount[32 * ((iter + 1) & 1)], total) : 0u;
        }
        __syncthreads();
        if (keep) {
            unsigned off = 0;
#pragma unroll
            for (int w = 0; w < SHADE_BLOCK / 64; w++) off += (w < wave) ? sCntB[w] : 0u;
            qOut[sBase + off + (unsigned)__popcll(mk & ltMask)] = i;
        }
    }
    // ---- 3. the segment (frag.glsl:823-879), then what a finished sample / a finished job leaves behind
    LobePending L; L.N = v3(0.0f); L.Pcr = 0.0f; L.w = 0; L.needG = false;
    if (live && EARLY) {
        shadeSegment<STK, TEX, FAST>(sc, fc, p, h.x, h.y, h.z, __float_as_int(h.w), st.G5, st.HX, i, L);      // (its answer is sampleDone's)
        if (sampleDone) {
            p.sum = p.sum + p.inc;                                 // col += trace(...)  (:910)
            p.sample++;
            // the sample loop goes on in this job's stream: chooseRay's six draws, made before trace() ended the sample, are behind it
            if (!jobDone && L.needG) p.rng = skipSixDraws(p.rng);
        }
    }
    if (jobDone) {
        float sr = fc.SAMPLE_RES;
        stS(b.colbuf + (size_t)(jobWords.x % b.ringFrames) * b.nSlots + jobWords.y, make_float4(p.sum.x / sr, p.sum.y / sr, p.sum.z / sr, 1.0f));   // col /= SAMPLE_RES (:915)
        if (gotJob) { startJob(b, fc, nextJob, p); needStart = true; newJob = true; }
        else p.alive = false;
    }
    // ---- 4. ONE site of randLambertianDistVec: a lane whose path goes on draws the Gaussian vector of its lobe (chooseRay, :775-804), a lane that starts a
    // sample the one of its lens jitter (frag.glsl:899-908) — a lane is one or the other; every store stays in slot order
    const bool drawLobe = live && !sampleDone && L.needG;
    vec3 G = v3(0.0f);
    if (needStart || drawLobe) G = randLambertianDistVec<FAST>(p.rng);
    if (needStart) { tracePrologue<STK>(p); cameraRayFrom<FAST>(fc, b.W, b.H, (int)(p.pix & 0xffffu), (int)(p.pix >> 16), G, p.O, p.D); }
    else if (drawLobe) p.D = lobeDirection<FAST>(L.w, G, L.N, p.D, 0.0f, L.Pcr);
    if (live) {
        {   // incLight leaves zero only at an emitter whose sample goes on, and returns to it with the next sample (tracePrologue): most segments
            // neither read nor write the group
            const bool nz = (__float_as_uint(p.inc.x) | __float_as_uint(p.inc.y) | __float_as_uint(p.inc.z)) != 0u;
            const bool changed = __float_as_uint(p.inc.x) != __float_as_uint(g3in.x) || __float_as_uint(p.inc.y) != __float_as_uint(g3in.y) ||
                                 __float_as_uint(p.inc.z) != __float_as_uint(g3in.z);
            if (DIRECT) { if (changed) stS(st.G3 + i, make_float4(p.inc.x, p.inc.y, p.inc.z, 0.0f)); p.incNZ = false; }
            else { if (nz && (changed || !incInMemory)) stS(st.G3 + i, make_float4(p.inc.x, p.inc.y, p.inc.z, 0.0f)); p.incNZ = nz; }
        }
        stS(st.G0 + i, make_float4(p.O.x, p.O.y, p.O.z, p.D.x));
        stS(st.G1 + i, make_float4(p.D.y, p.D.z, __uint_as_float(p.rng), __uint_as_float(packFlags(p))));
        stS(st.G2 + i, make_float4(p.col.x, p.col.y, p.col.z, __uint_as_float(p.sc0)));
        if (sampleDone) stS(st.G4 + i, make_float4(p.sum.x, p.sum.y, p.sum.z, __uint_as_float(p.pix)));
        if (newJob) st.J[i] = make_uint2(p.fi, p.ls);
        if (STK == 8) stS(st.S0 + i, make_float4(__uint_as_float(p.sc0), __uint_as_float(p.sc1), __uint_as_float(p.sc2), 0.0f));
        if (STK == 32) storeStack32(st, i, p);
        if (TRANS) { if (p.g5dirty || newJob) stS(st.G5 + i, make_float4(p.enter.x, p.enter.y, p.enter.z, p.dist)); }
    }
    if (STATS) {                                  // statistics (count mode only): one atomic per wave
        unsigned long long lm = __ballot(live && !isProbe);      // a thickness probe is part of the same directDiffuse call
        unsigned long long sm = __ballot(nSamp != 0);
        if (lm && lane == (__ffsll((long long)lm) - 1)) atomicAdd(&ctl->cnt[PT_CNT_SEGMENTS], (unsigned long long)__popcll(lm));
        if (sm && lane == (__ffsll((long long)sm) - 1)) atomicAdd(&ctl->cnt[PT_CNT_SAMPLES], (unsigned long long)__popcll(sm));
    }
}

// FRAME accumulation, frag.glsl:924-933, over one batch's frames in u_frameCount order (stream frames f0 .. f0+nFrames-1)
__global__ void __launch_bounds__(BLOCK) k_accumulate(Batch b, const FrameConst* fcp, float4* frame, unsigned f0, int nFrames, int firstFrame) {
    unsigned ls = blockIdx.x * BLOCK + threadIdx.x;
    if (ls >= (unsigned)b.nSlots) return;
    int gp;
    if (b.shardCount == 1) gp = (int)ls;
    else { if (ls >= (unsigned)b.nLocal) return; gp = b.pixList[ls]; }
    const FrameConst& fc = *fcp;
    if (inMouseOverlay(fc, gp % b.W, gp / b.W)) return;
    float4 F = frame[ls];
    for (int f = 0; f < nFrames; f++) {
        float4 c = ldS(b.colbuf + (size_t)((f0 + (unsigned)f) % b.ringFrames) * b.nSlots + ls);
        if ((float)(firstFrame + f) == 1.0f) F = make_float4(c.x, c.y, c.z, 1.0f);
        else F = make_float4(F.x + c.x, F.y + c.y, F.z + c.z, F.w + 1.0f);
    }
    frame[ls] = F;
}

// DEBUG != 0 (frag.glsl:916-918): the traversal heat-map of debugRayScene (:539-547) — no random numbers, no samples.  One thread per
// local pixel; every BVH is traversed on its own from closest_t = 1e30 with the un-offset ORIGIN and the un-normalised primary
// direction (exactly the thickness probe's rayBVH call), and rayBVH's .col (:534) is rebuilt from the traversal counters:
// boxTests = 2 per inner node popped, 0.1 per leaf popped, triTests never incremented.  The frames of the batch add the same colour.
__global__ void __launch_bounds__(64) k_debug_heatmap(DevScene sc, Batch b, const FrameConst* fcp, float4* frame, int firstFrame, int nFrames) {
    __shared__ int stk[64 * 64];
    unsigned ls = blockIdx.x * 64 + threadIdx.x;
    if (ls >= (unsigned)b.nLocal) return;
    const FrameConst& fc = *fcp;
    const unsigned xy = b.pixXY[ls];
    const int px = (int)(xy & 0xffffu), py = (int)(xy >> 16);
    uint32_t index;
    if (!pixelIndex(fc, b.W, b.H, px, py, index) || inMouseOverlay(fc, px, py)) return;
    float tcx = ((float)px + 0.5f) / (float)b.W, tcy = ((float)py + 0.5f) / (float)b.H;
    vec3 q = v3(((tcx * 2.0f - 1.0f) * -1.0f) * fc.screenSize, ((tcy * 2.0f - 1.0f) * fc.screenHratio) * fc.screenSize, fc.focalLength);
    const vec3 direction = vecmat(q, fc.camRot);                   // :894, not normalised
    const vec3 ORIGIN = v3(fc.origin[0], fc.origin[1], fc.origin[2]);
    vec3 ret = v3(0.0f);
    for (int ob = 0; ob < sc.numObj; ob++) {
        Counters c;
        float t, u, v; int prim;
        intersectScene<true>(sc, ORIGIN, direction, stk + threadIdx.x, 64, nullptr, nullptr, t, u, v, prim, c, true, ob);
        const int boxTests = (int)c.boxtests - 1;                  // without the root test of :468
        const int leaves = (int)c.nodes - boxTests / 2;
        float ocx = 0.0f;
        for (int k = 0; k < leaves; k++) ocx = ocx + 0.1f;
        const float cx = ocx * 0.1f + 0.0f + exp_(0.02f * (float)(0 - 150)), cz = 0.0f * 0.1f + exp_(0.01f * (float)(boxTests - 200)) + 0.0f;
        ret = ret + v3(cx / (float)sc.numObj, 0.0f / (float)sc.numObj, cz / (float)sc.numObj);
    }
    const unsigned slot = (b.shardCount == 1) ? (unsigned)(py * b.W + px) : ls;
    float4 F = frame[slot];
    for (int f = 0; f < nFrames; f++) {
        if ((float)(firstFrame + f) == 1.0f) F = make_float4(ret.x, ret.y, ret.z, 1.0f);
        else F = make_float4(F.x + ret.x, F.y + ret.y, F.z + ret.z, F.w + 1.0f);
    }
    frame[slot] = F;
}

// Which of the (up to 8) oldest batches that have not been accumulated yet does a live slot still work on?  ends.f[k] = first stream frame BEHIND the k-th of
// them (ascending): a live slot on frame f keeps the first batch with f < ends.f[k] busy.  One pass over the flags per host poll.
struct ScanEnds { unsigned f[8]; int n; };
// The host's look at the scheduler words: ONE wave copies Control into the group's snapshot in pinned, coherent host memory and, behind a system-scope fence, writes
// the group's number into the group's stamp.  The host reads both without a runtime call (no event, no copy engine, no synchronisation of the stream): this is the form
// that was verified under the checking allocator while a host heap corruption was being traced (profiles/r06_f_runtime_write_after_free.txt; its cause turned out to be
// hipStreamDestroy, see takeStream).
__global__ void __launch_bounds__(64) k_snapshot(const Control* ctl, Control* snap, volatile unsigned* stamp, unsigned seq) {
    const unsigned* s = reinterpret_cast<const unsigned*>(ctl);
    unsigned* d = reinterpret_cast<unsigned*>(snap);
    for (unsigned i = threadIdx.x; i < sizeof(Control) / 4; i += 64) d[i] = s[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *stamp = seq; __threadfence_system(); }
}
__global__ void __launch_bounds__(BLOCK) k_scan_inflight(State st, int nSlots, ScanEnds ends, Control* ctl) {
    unsigned i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (unsigned)nSlots) return;
    if (!(__float_as_uint(st.G1[i].w) & FL_ALIVE)) return;
    const unsigned f = st.J[i].x;
    if (f >= ends.f[ends.n - 1]) return;
    int k = 0;
    while (f >= ends.f[k]) k++;
    ctl->busy[k] = 1u;
}

// fragColor -> UNORM8 framebuffer -> glReadPixels(GL_RGB) -> Java signed-byte packing -> vertical flip (dispatch.java:804-833)
__global__ void __launch_bounds__(BLOCK) k_display(const float4* frame, int W, int H, float frameCount, int javaBytes, unsigned char* out) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= W * H) return;
    int x = i % W, y = i / W;
    float4 F = frame[i];
    float v[3] = {F.x / frameCount, F.y / frameCount, F.z / frameCount};
    int q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = (v[k] != v[k]) ? 0.0f : (v[k] < 0.0f ? 0.0f : (v[k] > 1.0f ? 1.0f : v[k]));
        q[k] = (int)__builtin_floorf(t * 255.0f + 0.5f);
    }
    int r = q[0], g = q[1], b = q[2];
    if (javaBytes) {
        int pix = (int)((unsigned)(int)(signed char)r << 16) + (int)((unsigned)(int)(signed char)g << 8) + (int)(signed char)b;
        r = (pix >> 16) & 0xff; g = (pix >> 8) & 0xff; b = pix & 0xff;
    }
    unsigned char* o = out + 3 * ((size_t)(H - 1 - y) * W + x);
    o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)b;
}

__global__ void k_unshard(const float4* gathered, const int* maps, int nSlots, int shardCount, float4* full) {
    size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= (size_t)nSlots * shardCount) return;
    int gp = maps[k];
    if (gp >= 0) full[gp] = gathered[k];
}

__global__ void k_init_control(Control* ctl) {                     // a new frame stream: job ids restart at 0
    ctl->nextJob = 0; ctl->jobEnd = 0; ctl->needRevive = 1; ctl->oldestBusy = 0; for (int k = 0; k < 8; k++) ctl->busy[k] = 0;
    for (int k = 0; k < 4; k++) ctl->exhausted[k] = 0;
    ctl->qCount[0] = 0; ctl->qCount[32] = 0;
}
// addJobs more jobs for the running stream.  If the pool already ran dry (a pull came back empty), the ids it overshot by
// were never started: hand them out again and let dead slots pull (k_revive); the tail queue is dropped, every slot is visited.
__global__ void k_submit(Control* ctl, unsigned addJobs, int streamStart, unsigned nSlots) {
    bool dry = false;
    for (int k = 0; k < 4; k++) { dry = dry || ctl->exhausted[k] != 0; ctl->exhausted[k] = 0; }
    if (ctl->nextJob > ctl->jobEnd) { ctl->nextJob = ctl->jobEnd; dry = true; }
    ctl->jobEnd += addJobs;
    ctl->needRevive = (dry || streamStart == 2) ? 1u : 0u;                              // 2: the pool has just grown by dead slots
    if (streamStart == 1) { ctl->needRevive = 2u; ctl->nextJob = min(nSlots, addJobs); }      // fresh pool: slot i starts job i, no pulling
    ctl->qCount[0] = 0; ctl->qCount[32] = 0;
}

__global__ void k_debug_math(int fn, const float* x, const float* y, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = sin_(x[i]); break;
        case 1: r = cos_(x[i]); break;
        case 2: r = log_(x[i]); break;
        case 3: r = exp_(x[i]); break;
        case 4: r = atan2_(x[i], y[i]); break;
        case 5: r = asin_(x[i]); break;
        // the RNG of frag.glsl:686-694 on the device (K1 vectors): x carries the uint32 state as bits
        case 6: { uint32_t st = __float_as_uint(x[i]); NextRandom(st); r = __uint_as_float(st); break; }                     // state after one call
        case 7: { uint32_t st = __float_as_uint(x[i]); r = __uint_as_float(NextRandom(st)); break; }                         // result of that call
        case 8: { uint32_t st = __float_as_uint(x[i]); r = random_(st); break; }                                             // random()
        case 9: r = unorm8((uint32_t)x[i]); break;                                                                           // byte / 255.0f as the samplers evaluate it
        default: r = __builtin_nanf("");
    }
    out[i] = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side

struct MultiCtx;
struct pt_ctx {
    MultiCtx* multi = nullptr;      // != nullptr: a multi-GPU group (pt_create_multi, pt_multi.hpp); everything below then lives in its per-device contexts
    int device = 0, W = 0, H = 0, shardRank = 0, shardCount = 1;
    hipStream_t ownStream = nullptr, stream = nullptr;
    // spatial partition (pt_set_option 21, an experiment: profiles/r06_c_cu_partition.txt): the intersect launches on a stream whose CU mask holds cuPartition eighths of
    // every XCD's CUs, the shading launches on the complement; events order extend(i) -> shade(i) -> extend(i+1), everything else stays on `stream`
    int cuPartition = 0, cuPartitionBuilt = 0; hipStream_t sExt = nullptr, sShade = nullptr; hipEvent_t evExt = nullptr, evShade = nullptr, evHost = nullptr;
    // raw SSBO contents (host copies, glBufferData semantics)
    std::vector<float> origin, rotation, mouse, tris, params, imp, ellip, bvhdata, mtl;
    std::vector<int32_t> bvhtree, leaftris, objidx;
    std::vector<uint8_t> sky; int skyW = 0, skyH = 0;
    struct HostTex { std::vector<uint8_t> rgba; int w = 0, h = 0; };
    std::vector<HostTex> textures;          // bindless table beyond the sky (index 0 mirrors `sky`)
    uchar4* dTexels = nullptr; TexRec* dTexTable = nullptr;      // all textures beyond the sky in one allocation + the bindless-style table
    bool sceneDirty = true, frameInDirty = true;
    bool trans = false, anySubsurface = false, ambiguousTriObj = false, anyMaps = false, ellipMaps = false; int* dTriObj = nullptr;
    int stackDepth = 1;
    // device scene
    float4 *dNodes = nullptr, *dTris = nullptr, *dShade = nullptr; ObjRoot* dRoots = nullptr; EllipRec* dEllip = nullptr; MatRec* dMats = nullptr;
    uchar4* dSky = nullptr; float* dNiTable = nullptr;
    int niBits = 0;                 // index-stack encoding of the path state: 0 no transmissive material (the stack is unobservable), 3 or 8 bits per slot, 32: the floats themselves
    unsigned char* dDisplay = nullptr;      // scratch of pt_read_display (W*H*3 bytes, allocated on first use)
    DevScene sc{};
    // shard
    std::vector<int32_t> pixList; int nLocal = 0, nSlotsImg = 0; int* dPixList = nullptr; unsigned* dPixXY = nullptr; int* dAllMaps = nullptr;
    static constexpr int IMAGES = 4;
    float4* dImage[IMAGES] = {nullptr, nullptr, nullptr, nullptr}; int curImage = 0;      // FRAME images (more than one only after pt_next_image)
    // path pool
    int poolSlots = 0;              // 0 = automatic: jobs/5 clamped to [2^20, 2^22] (enough rays per lane for the in-wave refill, short tail)
    int poolActive = 0; int allocSlots = 0; int allocNiBits = -1; bool allocHX = false;
    State st{};
    unsigned* dQueue[2] = {nullptr, nullptr};      // dense slot queues of the batch tail, by iteration parity
    float4* dColbuf = nullptr; int* dSeeds = nullptr; int ringFrames = 0;      // per-frame rings of the stream (Batch)
    // frame-stream scheduler (host view)
    struct Entry { unsigned jobEnd, f0; int nFrames, firstFrame, image; };      // a submitted, not yet accumulated batch
    std::deque<Entry> pending;
    FrameIn streamIn{};             // frame inputs the running stream was started with
    unsigned streamFrames = 0, streamJobs = 0, lastNextJob = 0, lastDelta = 0, launched = 0; int lastCheck = 24, iter = 0; bool draining = false;
    uint64_t lastSubmitJobs = 0, jobsThisImage = 0, jobsPerImage = 0;      // what the last submission added; jobs submitted for the current / the previous FRAME image
    FrameIn* dFrameIn = nullptr; FrameConst* dFc = nullptr; Control* dCtl = nullptr;
    // The host looks at the device's scheduler words once per GROUP of iterations: a group = its launches + (a scan of the oldest batches) + k_snapshot, which writes
    // Control into the group's pinned snapshot and then the group's number into the group's pinned STAMP.  Up to two groups are in flight: the host looks at a snapshot when
    // its stamp has arrived, so the stream always holds the next group's launches while one runs, and an asynchronous submission never waits for the iterations it started
    // (pump).
    struct Group { Control* h = nullptr; volatile unsigned* stamp = nullptr; unsigned seq = 0; int check = 0, iterEnd = 0, nScan = 0; unsigned scanF0 = 0, epoch = 0; int64_t predicted = 0; };
    Group grp[2]; int grpHead = 0, grpCount = 0; bool scanInFlight = false; unsigned submitEpoch = 0, groupSeq = 0;
    int64_t inflightPredicted = 0;  // jobs the groups in flight are expected to hand out (iterations x the rate of the last look): lastNextJob is as old as the oldest of them
    FrameIn* hFrameIn = nullptr; int32_t* hSeeds = nullptr;   // pinned staging (hSeeds: ring like dSeeds)
    // options / stats
    bool countStats = false, timing = false;
    int ldsBudget = 20 * 1024;
    int extendMode = 2;             // 0: one block per 256 lanes (k_extend), 1: persistent blocks (k_extend_persist), 2: the hand-written form of 1
                                    //    (pt_extend_gfx950.s) for the scenes it takes, 1 for the others
    float* dNodes80 = nullptr;      // node records of the hand-written kernel: the two references + (min pair, max pair, min pair) per axis (80 B), or + pad + (min pair, max pair) (64 B)
    int asmNodeStride = 80, asmNodeLayout = -1;      // bytes per record as built; pt_set_option 19: -1 automatic, 0 80-B, 1 64-B
    int asmGroupShift = 0; bool asmNoRootCull = false;      // more than 8 BVHs: log2 of the objects per group box of the per-ray cull; pt_set_option 20 switches the cull off (every group box infinite)
    void* dAsmDbg = nullptr;
    std::string asmError;           // a failed load / launch of the hand-written kernel (surfaces as PT_ERR_HIP from the render call)
    uint64_t asmLaunches = 0;
    bool asmEligible = false;       // this scene can run on the hand-written kernel (buildScene)
    std::string asmWhyNot;          // ... or why not (pt_debug: reported by option 12)
    // per block size (256, 1024, 512 threads) six code objects: 16-bit stack entries, Packed18, the same two with v_rcp_f32 (relaxed contract), 24-bit entries exact / relaxed
    hipModule_t asmModule[18] = {}; hipFunction_t asmFn[18] = {}; std::string asmLoadError[18];
    int asmTpb = 0;                 // threads per block of the hand-written kernel: 0 automatic (launchExtendAsm), 256, 1024
    bool debugExactExtend = false;  // pt_debug_intersect always probes the exact kernels
    int extendTpb = 256, extendCacheBytes = 8 * 1024, refillMin = 24, numCUs = 256;
    int noneMin = 8;                // lanes waiting for their next object / retirement that make that phase worth a trip
    bool noneMinSet = false;        // pt_set_option 3 was used
    int streamsOnDevice = 1;        // streams of the same multi-stream context on this context's GPU (pt_create_multi)
    bool extendCacheSet = false;    // pt_set_option 6 was used: the tile size is the caller's
    int forceNiBits8 = 0;           // pt_set_option 18 (tests): 1 = 8-bit index-stack codes even when the scene's dictionary fits 3 bits, 2 = the float stack
    bool fastContract = false, streamFast = false;      // the relaxed numeric contract (pt_set_option 16) as set / as the running stream was started with
    int asmLoop = -1;               // hand-written kernel's main loop: -1 automatic, 0 phase-voting, 1 fused trip (pt_set_option 14)
    int stackMode = 2, stackModeForce = -1;      // 0: short entries, 1: Packed18, 2: int (see k_extend_persist); Force: pt_set_option 11
    int pLdsNodes = 0, pLdsTris = 0; int extendMaxBlocksPerCU = 0; int innerKeepEighths = 6;
    int bfsNodes = 0x7fffffff;      // inner-node records kept in breadth-first order (whole levels); the rest follow depth-first (buildScene)
    uint64_t hostCnt[PT_CNT_N] = {0};
    struct KT { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; int64_t launches = 0; double ms = 0; std::vector<float> each; } kt[4];
};

namespace {

struct Scratch {            // device allocations of one call: released whichever way the call returns
    std::vector<void**> ptrs;
    ~Scratch() { for (void** p : ptrs) if (*p) { hipFree(*p); *p = nullptr; } }
};

int uploadVec(void** dptr, const void* src, size_t bytes, hipStream_t s) {
    if (*dptr) { HIP_TRY(hipFree(*dptr)); *dptr = nullptr; }
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc(dptr, bytes));
    if (src) HIP_TRY(hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

// tile-major enumeration of the pixels owned by `rank`
void shardPixels(int W, int H, int rank, int count, std::vector<int32_t>& out) {
    out.clear();
    int ntx = (W + TILE_W - 1) / TILE_W, nty = (H + TILE_H - 1) / TILE_H;
    for (int t = rank; t < ntx * nty; t += count) {
        int tx = t % ntx, ty = t / ntx;
        for (int y = ty * TILE_H; y < std::min(H, (ty + 1) * TILE_H); y++)
            for (int x = tx * TILE_W; x < std::min(W, (tx + 1) * TILE_W); x++) out.push_back(y * W + x);
    }
}
size_t shardSlots(int W, int H, int count) {
    size_t mx = 0; std::vector<int32_t> tmp;
    for (int r = 0; r < count; r++) { shardPixels(W, H, r, count, tmp); mx = std::max(mx, tmp.size()); }
    return (mx + BLOCK - 1) / BLOCK * BLOCK;
}

// Validates the reference's buffers and builds the device-private layout (see pt_device.hpp).
int buildScene(pt_ctx* c) {
    size_t nTris = c->tris.size() / 40, nNodes = c->bvhtree.size() / 3;
    if (c->params.size() < 12) return fail(PT_ERR_SCENE, "Parameters buffer (binding 4) must hold 12 floats");
    if (c->origin.size() < 3 || c->rotation.size() < 3) return fail(PT_ERR_SCENE, "ORIGIN/ROTATION (bindings 0,1) not set");
    if (c->mouse.size() < 3) return fail(PT_ERR_SCENE, "MOUSE_POS (binding 2) not set");
    if (c->mtl.empty()) return fail(PT_ERR_SCENE, "mtlData (binding 14) not set");
    if (c->objidx.empty()) return fail(PT_ERR_SCENE, "objIndices (binding 13) not set");
    // Implicit surfaces: the reference loops over them (frag.glsl:578-605) and rayImplicit returns 1e30 before anything else (:385-386), so `t < closest_t` never passes —
    // they are never hit and leave no trace in the image.  The buffer is accepted as the reference's scene code sends it (dispatch.java:429-456) and otherwise unread.
    if (c->imp.empty() || !(c->imp[0] >= 0.0f)) return fail(PT_ERR_SCENE, "ImpData (binding 5) not set: [count, fn x n, shift x 3n, scale x 3n, rot x 3n, mat x n]; send [0] for none");
    if (c->ellip.empty()) return fail(PT_ERR_SCENE, "EllipData (binding 7) not set");
    if (c->sky.empty()) return fail(PT_ERR_SCENE, "texture 0 (sky) not set");
    if (c->bvhdata.size() < 8 * nNodes) return fail(PT_ERR_SCENE, "BVHdata shorter than 8 floats per BVHtree node");
    // materials
    int me = (int)c->mtl[0];
    if (me < 48) return fail(PT_ERR_SCENE, "mtlData[0] (floats per material) must be >= 48");
    int nMat = (int)((c->mtl.size() - 1) / me);
    std::vector<MatRec> mats(std::max(nMat, 1));
    c->trans = false; c->anySubsurface = false; c->anyMaps = false;
    for (int m = 0; m < nMat; m++) {
        const float* F = c->mtl.data() + (size_t)me * m;      // F[k] == mtlData[me*m + k]
        MatRec& r = mats[m];
        // map_* slots of the 48-float record (dispatch.java:295-315): Ka 22, Kd 23, Ks 24, Pm 32, Pr 33, Pc 35, bump/norm 37, Tr 39, Ke 41
        // (map_Ps 34, map_Pcr 36, map_d 38, map_Ns 40 only change fields the render path never reads)
        r.map_Ka = (int)F[22]; r.map_Kd = (int)F[23]; r.map_Ks = (int)F[24]; r.map_Pm = (int)F[32]; r.map_Pr = (int)F[33]; r.map_Pc = (int)F[35];
        r.map_norm = (int)F[37]; r.map_Tr = (int)F[39]; r.map_Ke = (int)F[41];
        r.hasMaps = 0;
        for (int idx : {r.map_Ka, r.map_Kd, r.map_Ks, r.map_Ke, r.map_Tr, r.map_Pm, r.map_Pr, r.map_Pc, r.map_norm}) {
            if (idx <= -1) continue;
            r.hasMaps = 1;
            if ((size_t)idx >= c->textures.size() || c->textures[idx].rgba.empty())
                return fail(PT_ERR_SCENE, "a material names a texture index that was never uploaded with pt_set_texture");
        }
        for (int k = 0; k < 3; k++) { r.Kd[k] = F[4 + k]; r.Ks[k] = F[7 + k]; r.Tf[k] = F[13 + k]; r.Ke[k] = F[17 + k]; }
        r.Tr = F[12]; r.Ni = F[16]; r.Density = F[20]; r.illum = (int)F[21]; r.Pm = F[25]; r.Pr = F[26]; r.Pc = F[28]; r.Pcr = F[29]; r.subsurface = F[42];
        for (int k = 0; k < 3; k++) { r.Ka[k] = F[1 + k]; r.ssColor[k] = F[43 + k]; r.ssRadius[k] = F[46 + k]; }
        if (r.Tr > 0.0f || r.Tf[0] > 0.0f || r.illum == 5 || r.illum == 7 || r.map_Tr > -1) c->trans = true;   // (a Tr map can switch transmission on)
        if (r.subsurface > 0.0f) c->anySubsurface = true;
        if (r.hasMaps) c->anyMaps = true;
    }
    // the refraction-index dictionary (pt_device.hpp, DevScene::ni8): 0.0f, 1.0029f, then every distinct Ni bit pattern among the materials
    std::vector<float> niDict = {0.0f, 1.0029f};
    for (int m = 0; m < nMat; m++) {
        uint32_t bits; std::memcpy(&bits, &mats[m].Ni, 4);
        int code = -1;
        for (size_t k = 0; k < niDict.size(); k++) { uint32_t kb; std::memcpy(&kb, &niDict[k], 4); if (kb == bits) { code = (int)k; break; } }
        if (code < 0) { code = (int)niDict.size(); niDict.push_back(mats[m].Ni); }
        mats[m].niCode = code;
    }
    // (more values than the 8-bit dictionary holds, 0.0 and 1.0029 included: the path state carries the ten floats of the shader's stack themselves, frag.glsl:136-158)
    c->niBits = !c->trans ? 0 : (niDict.size() > 256 || c->forceNiBits8 == 2) ? 32 : ((niDict.size() <= 8 && !c->forceNiBits8) ? 3 : 8);
    if (niDict.size() > 256) { niDict.resize(256); for (auto& m : mats) if (m.niCode > 255) m.niCode = 0; }      // (the codes are not read in that form)
    niDict.resize(std::max<size_t>(niDict.size(), 8), 0.0f);
    // objects / BVH
    int numObj = c->objidx[0];
    if (numObj < 0 || (size_t)numObj + 1 > c->objidx.size()) return fail(PT_ERR_SCENE, "objIndices[0] exceeds the buffer");
    auto childOf = [&](int n, int side) { return c->bvhtree[3 * (size_t)n + 1 + side]; };
    std::vector<int> newIdx(nNodes, -1), depth(nNodes, 0), objOf(nNodes, -1);
    std::vector<int> order;                                     // inner nodes in multi-root BFS order
    std::vector<char> seen(nNodes, 0);
    std::vector<int> frontier;
    auto isLeaf = [&](int n) { return (childOf(n, 0) | childOf(n, 1)) == -1; };   // bitwise OR, frag.glsl:478
    for (int o = 0; o < numObj; o++) {
        int r = c->objidx[1 + o];
        if (r < 0 || (size_t)r >= nNodes) return fail(PT_ERR_SCENE, "objIndices root out of range");
        if (seen[r]) return fail(PT_ERR_SCENE, "BVH node reachable twice (not a tree)");
        seen[r] = 1; frontier.push_back(r); objOf[r] = o;
    }
    int maxInnerDepth = -1;
    {
        std::vector<int> cur = frontier, nxt;
        int d = 0;
        while (!cur.empty()) {
            nxt.clear();
            for (int n : cur) {
                depth[n] = d;
                if (isLeaf(n)) continue;
                maxInnerDepth = std::max(maxInnerDepth, d);
                newIdx[n] = (int)order.size(); order.push_back(n);
                for (int s = 0; s < 2; s++) {
                    int ch = childOf(n, s);
                    if (ch < 0 || (size_t)ch >= nNodes) return fail(PT_ERR_SCENE, "BVHtree child index out of range");
                    if (seen[ch]) return fail(PT_ERR_SCENE, "BVH node reachable twice (not a tree)");
                    seen[ch] = 1; nxt.push_back(ch); objOf[ch] = objOf[n];
                }
            }
            cur.swap(nxt); d++;
        }
    }
    // Below the top levels (the LDS tile and what every XCD's L2 keeps hot anyway) the records are laid out in depth-first order
    // instead: a node and its left child are then neighbours, and a subtree's last levels share a few cache lines — a big tree's
    // deep fetches are what misses L2.  Only the addresses change; which nodes a ray visits, and in which order, does not.
    if ((size_t)c->bfsNodes < order.size()) {
        int cut = -1; size_t upTo = 0;                          // deepest level that is still completely inside the BFS prefix
        for (size_t k = 0; k < order.size(); k++) {
            if (k + 1 == order.size() || depth[order[k + 1]] != depth[order[k]]) {
                if (k + 1 <= (size_t)c->bfsNodes) { cut = depth[order[k]]; upTo = k + 1; } else break;
            }
        }
        std::vector<int> reordered(order.begin(), order.begin() + upTo), stack;
        for (size_t k = upTo; k < order.size() && depth[order[k]] == cut + 1; k++) {
            stack.assign(1, order[k]);
            while (!stack.empty()) {
                int n = stack.back(); stack.pop_back();
                reordered.push_back(n);
                int L = childOf(n, 0), R = childOf(n, 1);
                if (!isLeaf(R)) stack.push_back(R);
                if (!isLeaf(L)) stack.push_back(L);
            }
        }
        if (reordered.size() != order.size()) return fail(PT_ERR_SCENE, "internal: depth-first relayout lost nodes");
        order.swap(reordered);
        for (size_t k = 0; k < order.size(); k++) newIdx[order[k]] = (int)k;
    }
    int need = maxInnerDepth + 2;                               // worst-case entries on rayBVH's stack
    if (need > 64) return fail(PT_ERR_SCENE, "BVH too deep for the reference's `int stack[64]` (frag.glsl:465)");
    c->stackDepth = std::max(need, 1);
    // leaf-ordered triangle records
    std::vector<float4> triRecs; std::vector<int> leafRef(nNodes, REF_EMPTY);
    std::vector<int> triObj(std::max<size_t>(nTris, 1), -1);      // triangle -> object whose BVH holds it (hit.parentID of frag.glsl:573)
    c->ambiguousTriObj = false;
    auto f4 = [](float a, float b, float cc, float d) { return make_float4(a, b, cc, d); };
    auto asf = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    for (size_t n = 0; n < nNodes; n++) {
        if (!seen[n] || !isLeaf((int)n)) continue;
        int s = (int)c->bvhdata[8 * n + 6], e = (int)c->bvhdata[8 * n + 7];
        if (e <= s) continue;                                   // empty leaf
        if (s < 0 || (size_t)e > c->leaftris.size()) return fail(PT_ERR_SCENE, "leaf index range outside leafTriIndices");
        leafRef[n] = -((int)(triRecs.size() / 3) + 1);
        for (int i = s; i < e; i++) {
            int t = c->leaftris[i];
            if (t < 0 || (size_t)t >= nTris) return fail(PT_ERR_SCENE, "leafTriIndices entry outside the triangle buffer");
            const float* T = c->tris.data() + 40 * (size_t)t;
            int mat = (int)T[36];
            if (mat < 0 || mat >= nMat) return fail(PT_ERR_SCENE, "triangle material index out of range (SURVEY.md Q-14: OBJ faces before any o/g line get -1)");
            if (triObj[t] == -1) triObj[t] = objOf[n]; else if (triObj[t] != objOf[n]) { triObj[t] = -2; c->ambiguousTriObj = true; }
            float e1x = T[4] - T[0], e1y = T[5] - T[1], e1z = T[6] - T[2], e2x = T[8] - T[0], e2y = T[9] - T[1], e2z = T[10] - T[2];
            uint32_t idl = (uint32_t)t | (i == e - 1 ? 0x80000000u : 0u);
            triRecs.push_back(f4(T[0], T[1], T[2], e1x)); triRecs.push_back(f4(e1y, e1z, e2x, e2y)); triRecs.push_back(f4(e2z, asf(idl), 0, 0));
        }
    }
    auto refOf = [&](int n) { return isLeaf(n) ? leafRef[n] : newIdx[n]; };
    std::vector<float4> nodeRecs;
    for (int n : order) {
        int L = childOf(n, 0), R = childOf(n, 1);
        const float* A = c->bvhdata.data() + 8 * (size_t)L; const float* B = c->bvhdata.data() + 8 * (size_t)R;
        nodeRecs.push_back(f4(A[0], B[0], A[1], B[1])); nodeRecs.push_back(f4(A[2], B[2], A[3], B[3])); nodeRecs.push_back(f4(A[4], B[4], A[5], B[5]));
        nodeRecs.push_back(f4(asf((uint32_t)refOf(L)), asf((uint32_t)refOf(R)), 0, 0));
    }
    // The hand-written intersect kernel (pt_extend_gfx950.s) reads its own node records.  80 B: the two references, then per axis (Lmin, Rmin | Lmax,
    // Rmax | Lmin, Rmin), so that a lane whose direction component is negative starts 8 B further in and receives (near pair, far pair) without a
    // min / max.  Trees that do not fit the caches pay for those bytes on every node visit (C4: 552 B per segment, the chip at 0.61 of its HBM peak):
    // they get 64-B records — references, pad, (Lmin, Rmin | Lmax, Rmax) per axis — and the kernel's min/max step (pt_set_option 19 overrides).
    const size_t nInner = order.size();
    // (what decides is whether the records the rays walk through fit an XCD's L2 beside the state stream: C4's single 100 k-node tree gains 5 % from the small
    //  records — and so do C6's 64 trees of 1.5 k nodes, 7.8 MB in all, +6.2 %, since the per-ray cull of the object loop took the 64 root tests per ray out of its
    //  vector instructions; round 4, before the cull, measured -2 % there and chose by the largest tree: profiles/r04_d_node_record_layout.txt, r05_f_*)
    const int asmStride = c->asmNodeLayout == 0 ? 80 : c->asmNodeLayout == 1 ? 64 : (nInner * 80 > (size_t)ASM_NODES_80B_LIMIT ? 64 : 80);
    const int W_ = asmStride / 4;
    std::vector<float> nodes80(std::max<size_t>(nInner, 1) * W_ + 40, 0.0f);      // (+ 160 B: developer builds of the kernel read behind a record, -DFETCH_EXTRA)
    bool boxesOrdered = true, anyEmpty = false;
    for (size_t k = 0; k < nInner; k++) {
        const int n = order[k], L = childOf(n, 0), R = childOf(n, 1);
        const float* A = c->bvhdata.data() + 8 * (size_t)L; const float* B = c->bvhdata.data() + 8 * (size_t)R;
        float* o = nodes80.data() + (size_t)W_ * k;
        for (int ax = 0; ax < 3; ax++) {
            float* g = asmStride == 80 ? o + 2 + 6 * ax : o + 4 + 4 * ax;
            g[0] = A[ax]; g[1] = B[ax]; g[2] = A[3 + ax]; g[3] = B[3 + ax];
            if (asmStride == 80) { g[4] = A[ax]; g[5] = B[ax]; }
            if (!(A[ax] <= A[3 + ax]) || !(B[ax] <= B[3 + ax])) boxesOrdered = false;      // min > max or a NaN: only the min/max form of rayBox is right
        }
        const int lr = refOf(L), rr = refOf(R);
        std::memcpy(&o[0], &lr, 4); std::memcpy(&o[1], &rr, 4);
        if (lr == REF_EMPTY || rr == REF_EMPTY) anyEmpty = true;
    }
    c->asmNodeStride = asmStride;
    std::vector<ObjRoot> roots(std::max(numObj, 8));           // (the hand-written kernel fetches root records in batches of four: at least eight exist)
    for (int o = 0; o < numObj; o++) {
        int r = c->objidx[1 + o]; const float* A = c->bvhdata.data() + 8 * (size_t)r;
        for (int k = 0; k < 3; k++) { roots[o].bmin[k] = A[k]; roots[o].bmax[k] = A[3 + k]; }
        roots[o].ref = refOf(r); roots[o].pad = 0;
        // an empty root leaf would be "visited" by the reference and find nothing: it can simply never be pushed
    }
    // More than 8 BVHs: the hand-written kernel culls the object loop (frag.glsl:563-577) per ray with ONE pass over at most 64 GROUP boxes at refill
    // (pt_extend_gfx950.s, .Lmask_loop): group g = objects [g << s, (g + 1) << s), its box the union of their root boxes, appended to the root records.  A ray
    // that misses a group's box misses every root box in it (boxes of subsets; IEEE subtraction and multiplication are monotone, the ray regular: finite origin,
    // finite non-zero reciprocal direction), and a BVH whose root box the ray misses contributes nothing: rayBVH would pop the root, find neither child box hit
    // (children lie inside the root box) and return (:468-472, :521-531) — PROVIDED the root is an inner node whose child boxes lie inside an ordered root box.
    // A group holding a root that does not promise this (a leaf root: its triangles are tested whatever the box says, :478-520; foreign buffers whose children
    // stick out) gets the box (-inf, +inf): never culled.  Irregular rays skip the cull in the kernel.
    c->asmGroupShift = 0;
    if (numObj > 8) {
        int sft = 0;
        while (((numObj + (1 << sft) - 1) >> sft) > 64) sft++;
        c->asmGroupShift = sft;
        const int nGroups = (numObj + (1 << sft) - 1) >> sft;
        const float inf = std::numeric_limits<float>::infinity();
        std::vector<ObjRoot> groups(64);
        for (int g = 0; g < 64; g++) { for (int k = 0; k < 3; k++) { groups[g].bmin[k] = inf; groups[g].bmax[k] = -inf; } groups[g].ref = 0; groups[g].pad = 0; }
        for (int o = 0; o < numObj; o++) {
            const int r = c->objidx[1 + o]; const float* A = c->bvhdata.data() + 8 * (size_t)r;
            bool cullable = !isLeaf(r) && !c->asmNoRootCull;
            for (int k = 0; k < 3 && cullable; k++) {
                if (!(A[k] <= A[3 + k])) cullable = false;                                      // ordered, no NaN
                for (int side = 0; side < 2 && cullable; side++) {
                    const float* Ch = c->bvhdata.data() + 8 * (size_t)childOf(r, side);
                    // BOTH planes of the child inside the root's range: rayBox takes min / max of the two plane distances (:412-413), so an inverted child
                    // (min > max) whose `max` lies below the root's min would stick out of the root although its `min` and `max` each pass a one-sided test (NaN: not)
                    if (!(Ch[k] >= A[k] && Ch[k] <= A[3 + k] && Ch[3 + k] >= A[k] && Ch[3 + k] <= A[3 + k])) cullable = false;
                }
            }
            ObjRoot& G = groups[o >> sft];
            if (!cullable) G.pad = 1;
            for (int k = 0; k < 3; k++) { G.bmin[k] = std::min(G.bmin[k], A[k]); G.bmax[k] = std::max(G.bmax[k], A[3 + k]); }
        }
        for (int g = 0; g < nGroups; g++) if (groups[g].pad) { for (int k = 0; k < 3; k++) { groups[g].bmin[k] = -inf; groups[g].bmax[k] = inf; } }
        roots.insert(roots.end(), groups.begin(), groups.end());          // at roots[numObj .. numObj + 64)
    }
    std::vector<float4> shade(std::max<size_t>(nTris, 1) * 4);
    for (size_t t = 0; t < nTris; t++) {
        const float* T = c->tris.data() + 40 * t;
        shade[4 * t] = f4(T[12], T[13], T[14], T[16]); shade[4 * t + 1] = f4(T[17], T[18], T[24], T[25]);
        shade[4 * t + 2] = f4(T[28], T[29], T[32], asf((uint32_t)(int)T[36])); shade[4 * t + 3] = f4(T[33], 0, 0, 0);
    }
    // ellipsoids (frag.glsl:606-611 layout)
    int nE = (int)c->ellip[0];
    if (nE < 0 || c->ellip.size() < (size_t)1 + 11 * (size_t)nE) return fail(PT_ERR_SCENE, "EllipData shorter than its count says");
    std::vector<EllipRec> er(std::max(nE, 1));
    bool ellipMaps = false;
    for (int i = 0; i < nE; i++) {
        const float* E = c->ellip.data();
        EllipRec& r = er[i]; std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) { r.c[k] = E[1 + 3 * i + k]; r.st[k] = E[1 + nE * 3 + 3 * i + k]; r.rot[k] = E[1 + nE * 6 + 3 * i + k]; }
        r.r = E[1 + nE * 9 + i]; r.mat = (int)E[1 + nE * 10 + i];
        if (r.mat < 0 || r.mat >= nMat) return fail(PT_ERR_SCENE, "ellipsoid material index out of range");
        if (mats[r.mat].hasMaps) ellipMaps = true;               // sampled at the uv of the closest triangle found before the ellipsoid (frag.glsl:574 vs :619-630): State::HX
    }
    c->ellipMaps = ellipMaps;
    // upload
    hipStream_t s = c->stream;
    HIP_TRY(hipStreamSynchronize(s));
    int rc;
    if ((rc = uploadVec((void**)&c->dNodes, nodeRecs.data(), nodeRecs.size() * 16, s))) return rc;
    if ((rc = uploadVec((void**)&c->dNodes80, nodes80.data(), nodes80.size() * 4, s))) return rc;
    if ((rc = uploadVec((void**)&c->dTris, triRecs.data(), triRecs.size() * 16, s))) return rc;
    if ((rc = uploadVec((void**)&c->dShade, shade.data(), shade.size() * 16, s))) return rc;
    if ((rc = uploadVec((void**)&c->dTriObj, triObj.data(), triObj.size() * 4, s))) return rc;
    if ((rc = uploadVec((void**)&c->dRoots, roots.data(), roots.size() * sizeof(ObjRoot), s))) return rc;
    if ((rc = uploadVec((void**)&c->dEllip, er.data(), er.size() * sizeof(EllipRec), s))) return rc;
    if ((rc = uploadVec((void**)&c->dMats, mats.data(), mats.size() * sizeof(MatRec), s))) return rc;
    // textures stay the RGBA8 texels the caller uploaded (dispatch.java:349-354: GL_RGBA8); byte / 255.0f happens at fetch (unorm8, pt_device.hpp)
    if ((rc = uploadVec((void**)&c->dSky, c->sky.data(), (size_t)c->skyW * c->skyH * 4, s))) return rc;
    if ((rc = uploadVec((void**)&c->dNiTable, niDict.data(), niDict.size() * 4, s))) return rc;
    // the texture table beyond the sky: ONE allocation and one asynchronous copy for all textures
    std::vector<TexRec> table(std::max<size_t>(c->textures.size(), 1));
    table[0].data = c->dSky; table[0].w = c->skyW; table[0].h = c->skyH;
    std::vector<uint8_t> texels; std::vector<size_t> texOff(table.size(), 0);
    for (size_t ti = 1; ti < c->textures.size(); ti++) {
        const pt_ctx::HostTex& T = c->textures[ti];
        table[ti].data = nullptr; table[ti].w = T.w; table[ti].h = T.h;
        if (T.rgba.empty()) continue;
        texOff[ti] = texels.size() / 4;
        texels.insert(texels.end(), T.rgba.begin(), T.rgba.begin() + (size_t)T.w * T.h * 4);
    }
    if ((rc = uploadVec((void**)&c->dTexels, texels.data(), texels.size(), s))) return rc;
    for (size_t ti = 1; ti < c->textures.size(); ti++) if (!c->textures[ti].rgba.empty()) table[ti].data = c->dTexels + texOff[ti];
    if ((rc = uploadVec((void**)&c->dTexTable, table.data(), table.size() * sizeof(TexRec), s))) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    DevScene& sc = c->sc;
    sc.nodes = c->dNodes; sc.nNodes = (int)order.size(); sc.tris = c->dTris; sc.nTriRecs = (int)(triRecs.size() / 3);
    sc.shade = c->dShade; sc.nTris = (int)nTris; sc.triObj = c->dTriObj; sc.roots = c->dRoots; sc.numObj = numObj; sc.ellip = c->dEllip; sc.numEllip = nE;
    for (int k = 0; k < 8; k++) sc.ni8[k] = niDict[k];
    sc.niTable = c->dNiTable;
    sc.mats = c->dMats; sc.numMat = nMat; sc.sky = c->dSky; sc.skyW = c->skyW; sc.skyH = c->skyH; sc.tex = c->dTexTable; sc.numTex = (int)table.size();
    // LDS tile: as many leading (top-of-tree) node records and triangle records as the budget allows
    int budget = c->ldsBudget - c->stackDepth * BLOCK * 4;
    int ln = 0, lt = 0;
    if (budget > 0) {
        ln = std::min(sc.nNodes, budget / 64);
        int rest = budget - ln * 64;
        lt = std::min(sc.nTriRecs, rest / 48);
        if (ln < sc.nNodes) lt = std::min(lt, 0);               // triangles only once every node fits
    }
    sc.ldsNodes = ln; sc.ldsTris = lt;
    // persistent kernel: one staged tile per resident block
    c->stackMode = (sc.nNodes < 32767 && sc.nTriRecs < 32767) ? 0 : (sc.nNodes <= 131071 && sc.nTriRecs <= 131071 && c->stackDepth <= 32) ? 1 : 2;
    if (c->stackModeForce >= 0) c->stackMode = std::max(c->stackMode, c->stackModeForce);      // only ever towards wider entries
    {
        int cb = c->extendCacheBytes;
        int pn = std::min(sc.nNodes, cb / 64);
        int pt_ = (pn == sc.nNodes) ? std::min(sc.nTriRecs, (cb - pn * 64) / 48) : 0;
        c->pLdsNodes = pn; c->pLdsTris = pt_;
    }
    // which scenes the hand-written kernel takes (the others run on the compiled k_extend_persist, same results)
    for (int o = 0; o < numObj; o++) if (roots[o].ref == REF_EMPTY) anyEmpty = true;
    c->asmWhyNot.clear();
    if (numObj < 1 || numObj > 1024) c->asmWhyNot = "no BVH or more than 1024";
    else if (anyEmpty) c->asmWhyNot = "a leaf without triangles";
    else if (!boxesOrdered && asmStride == 80) c->asmWhyNot = "a node box with min > max or a NaN";      // (the 64-B records' min/max step is rayBox as written)
    else if (triRecs.size() / 3 >= (1u << 23) - 1 || order.size() >= (1u << 23) - 1) c->asmWhyNot = "more than 2^23 - 2 inner nodes or triangle records (24-bit stack entries)";
    c->asmEligible = c->asmWhyNot.empty();
    c->sceneDirty = false;
    return 0;
}

int ensurePool(pt_ctx* c, int capacity) {               // capacity >= poolActive: room for a pool that grows while a stream runs
    capacity = std::max(capacity, c->poolActive);
    if (c->allocSlots >= capacity && c->allocNiBits == c->niBits && c->allocHX == c->ellipMaps) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->allocSlots = 0;                                            // until every allocation below has succeeded there is no pool
    float4** groups[] = {&c->st.G0, &c->st.G1, &c->st.G2, &c->st.G3, &c->st.G4, &c->st.H, &c->st.G5, &c->st.S0, &c->st.HX};
    for (auto g : groups) if (*g) { HIP_TRY(hipFree(*g)); *g = nullptr; }
    if (c->st.J) { HIP_TRY(hipFree(c->st.J)); c->st.J = nullptr; }
    for (unsigned** q : {&c->dQueue[0], &c->dQueue[1]}) if (*q) { HIP_TRY(hipFree(*q)); *q = nullptr; }
    size_t n = (size_t)capacity;
    for (int k = 0; k < 9; k++) {
        if ((k == 6 && c->niBits == 0) || (k == 7 && c->niBits < 8) || (k == 8 && !c->ellipMaps)) continue;      // G5: transmissive scenes; S0: 8-bit index-stack codes, or three planes of floats; HX: mapped ellipsoids
        HIP_TRY(hipMalloc((void**)groups[k], n * 16 * ((k == 7 && c->niBits == 32) ? 3 : 1)));
    }
    HIP_TRY(hipMalloc((void**)&c->st.J, n * 8));
    HIP_TRY(hipMalloc((void**)&c->dQueue[0], n * 4));
    HIP_TRY(hipMalloc((void**)&c->dQueue[1], n * 4));
    c->allocSlots = capacity; c->allocNiBits = c->niBits; c->allocHX = c->ellipMaps; c->st.s0Plane = (unsigned)capacity;
    return 0;
}

int nextEventPair(pt_ctx::KT& k) {
    if (k.used == k.ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        k.ev.emplace_back(a, b);
    }
    return (int)k.used++;
}
#define TIMED_LAUNCH_ON(strm, kidx, ...)                                    \
    do {                                                                   \
        int ev_ = c->timing ? nextEventPair(c->kt[kidx]) : -1;            \
        if (ev_ >= 0) hipEventRecord(c->kt[kidx].ev[ev_].first, strm);     \
        __VA_ARGS__;                                                       \
        if (ev_ >= 0) hipEventRecord(c->kt[kidx].ev[ev_].second, strm);    \
    } while (0)
#define TIMED_LAUNCH(kidx, ...) TIMED_LAUNCH_ON(s, kidx, __VA_ARGS__)

struct PoolRun {            // host view of the path pool while a batch runs
    hipStream_t stream; State st; unsigned launched; int iter;      // launched: the host's upper bound on the slots an iteration visits
};
template <bool COUNT, typename StackT, int TPB>
void launchEP(pt_ctx* c, const PoolRun& pr, const DevScene& sc, size_t lds, int grid) {
    int nObjLds = std::min(sc.numObj, 8);
    // the two rare features of the kernel are compiled into a variant of their own (their registers cost the common one spills):
    // thickness probes (RAYTRACING == 0 of the running stream, not of a later upload) and the side record of mapped ellipsoids
    if (c->streamIn.params[9] != 1.0f || c->ellipMaps) {
        hipLaunchKernelGGL((k_extend_persist<COUNT, StackT, TPB, true>), dim3(grid), dim3(TPB), lds, pr.stream, sc, pr.st, c->dQueue[pr.iter & 1], pr.iter, (int)pr.launched, c->dCtl, c->refillMin,
                           c->innerKeepEighths, nObjLds, c->noneMin);
        return;
    }
    hipLaunchKernelGGL((k_extend_persist<COUNT, StackT, TPB, false>), dim3(grid), dim3(TPB), lds, pr.stream, sc, pr.st, c->dQueue[pr.iter & 1], pr.iter, (int)pr.launched, c->dCtl, c->refillMin,
                       c->innerKeepEighths, nObjLds, c->noneMin);
}
// Kernel arguments of pt_extend_gfx950.s (offsets are written into the assembly)
struct EpAsmArgs {
    const void *nodes80, *tris, *roots, *G0, *G1; void* H; const unsigned* queue; Control* ctl;
    int ldsNodes, ldsTris, numObj, iter, nSlots, refillMin, keepEighths, noneMin;
    unsigned divM, divS, nWaves, mode;      // mode: 1 the fused trip, 0 the phase-voting loop
    void* dbg;                      // developer builds of the assembly (-DPT_ASM_DEBUG): 32 B per wave
    const void* ellip; int numEllip, nodeStride;      // EllipRec array (rotation matrices by k_frame_setup); bytes per node record (80 or 64)
    int groupShift, stackDepth;     // more than 8 BVHs: log2 of the objects per group box (the 64 group boxes follow the root records); levels of the traversal stacks (24-bit entries: where the byte array starts)
    void* HX;                       // State::HX of scenes whose ellipsoids carry texture-mapped materials, else null
};
static_assert(sizeof(EpAsmArgs) == 152 && offsetof(EpAsmArgs, HX) == 144 && offsetof(EpAsmArgs, ellip) == 120 && offsetof(EpAsmArgs, groupShift) == 136, "EpAsmArgs layout is part of the assembly");
static_assert(sizeof(EllipRec) == 128 && offsetof(EllipRec, rotated) == 32 && offsetof(EllipRec, R) == 48, "EllipRec layout is part of the assembly");
#ifndef PT_EXTEND_INC
#define PT_EXTEND_INC "pt_extend_hsaco.inc"
#endif
#include PT_EXTEND_INC              // the assembled code objects (build.py): pt_extend_hsaco_s16[], pt_extend_hsaco_p18[]

// One code object per variant, loaded when a launch first needs it (a stream uses one or two of the eight).  A variant that failed to load stays
// failed: the error is latched, nothing half-loaded is kept and later launches do not retry.
int loadAsmKernel(pt_ctx* c, int k) {
    if (c->asmFn[k]) return 0;
    if (!c->asmLoadError[k].empty()) return fail(PT_ERR_HIP, c->asmLoadError[k]);
    const void* images[18] = {pt_extend_hsaco_s16, pt_extend_hsaco_p18, pt_extend_hsaco_s16f, pt_extend_hsaco_p18f, pt_extend_hsaco_p24, pt_extend_hsaco_p24f,
                              pt_extend_hsaco_s16w, pt_extend_hsaco_p18w, pt_extend_hsaco_s16fw, pt_extend_hsaco_p18fw, pt_extend_hsaco_p24w, pt_extend_hsaco_p24fw,
                              pt_extend_hsaco_s16h, pt_extend_hsaco_p18h, pt_extend_hsaco_s16fh, pt_extend_hsaco_p18fh, pt_extend_hsaco_p24h, pt_extend_hsaco_p24fh};
    hipModule_t m = nullptr; hipFunction_t f = nullptr;
    hipError_t e = hipModuleLoadData(&m, images[k]);
    if (e == hipSuccess) e = hipModuleGetFunction(&f, m, "pt_extend_asm");
    if (e != hipSuccess) {
        if (m) hipModuleUnload(m);
        c->asmLoadError[k] = std::string("code object ") + std::to_string(k) + " of pt_extend_asm: " + hipGetErrorString(e);
        return fail(PT_ERR_HIP, c->asmLoadError[k]);
    }
    c->asmModule[k] = m; c->asmFn[k] = f;
    return 0;
}

// true: launched.  false: this launch is not one the hand-written kernel takes (the caller uses the compiled kernel)
bool launchExtendAsm(pt_ctx* c, const PoolRun& pr) {
    // (RAYTRACING == 0, directDiffuse: its rays are ordinary rayScene calls, and the thickness probes of subsurface materials — FL_PROBE: no offset, one BVH, no
    //  ellipsoids — are set up at the kernel's refill)
    if (!c->asmEligible || c->countStats || c->extendTpb != 256) return false;
    const DevScene& sc = c->sc;
    // Block size.  What bounds this kernel is its CU's instruction issue and vector-memory pipe together (profiles/r03_h_*): node steps served from
    // the LDS tile cost neither a tag lookup nor a round trip, and the tile is per BLOCK — the same bytes eight times over with 256-thread blocks.
    // Alone on its GPU the kernel therefore runs 2 blocks of 1024 threads per CU over a 32 KB tile instead of 8 x 256 over 8 KB (C3 +6.6 %, C4 +10 %,
    // C5 +7 %, C2 +-0).  When the context's streams share the GPU the small blocks win (their slots free one by one for the other stream's
    // shading blocks: 1024-thread blocks -4...7 %), and so they do for a launch too small to give every CU its two large blocks.
    const bool part = c->sExt != nullptr && pr.stream == c->sExt;      // spatial partition: the kernel has its CUs to itself, but only those
    const int cus = part ? c->numCUs * c->cuPartitionBuilt / 8 : c->numCUs;
    const bool sharedGpu = c->streamsOnDevice > 1 && !part;
    const bool lazyRoots = sc.numObj > 8;                      // more than 8 BVHs: root records in LDS, tested when a BVH's turn comes (no per-lane distances)
    const size_t entryBytes = c->stackMode == 2 ? 3 : 2;       // traversal-stack entry in LDS: 16 bits (+ 2 in registers: Packed18), or 16 + 8 (24-bit entries)
    const size_t perLane = (lazyRoots ? 0 : (size_t)sc.numObj * 4) + (size_t)c->stackDepth * entryBytes;      // root-box distances + traversal stack of one lane
    const size_t rootBytes = lazyRoots ? ((size_t)sc.numObj + 64) * 32 : 0;      // the root records and the 64 group boxes of the per-ray cull (buildScene)
    const bool largeFits = 2 * (perLane * 1024 + rootBytes + 48 + 16384) <= (size_t)160 * 1024;      // two large blocks per CU with at least a 16 KB tile each (deep trees: stacks)
    const int TPB = c->asmTpb ? c->asmTpb : (!sharedGpu && c->streamsOnDevice == 1 && largeFits && pr.launched >= (uint64_t)cus * 2048 ? 1024 : 256);
    const int BPW = TPB / 256;                                  // how many 256-thread blocks one block stands for
    const size_t fixed = (lazyRoots ? rootBytes : (size_t)sc.numObj * 4 * TPB) + 48 + (size_t)c->stackDepth * entryBytes * TPB;      // root-box distances (or root records), root references + ray cursor, traversal stacks
    const size_t ldsPerCU = 160 * 1024;                        // gfx950; one block may take all of it
    if (fixed + 2048 > ldsPerCU) return false;
    // Blocks per CU and tile: alone on the GPU the kernel wants every wave slot (8 blocks of 256 threads, 8 KB tile).  When the context's streams
    // share the GPU (pt_create_multi with a device listed more than once) 6 blocks with a 16 KB tile are worth more: the two slots per SIMD it
    // leaves let the other stream's shading blocks run beside it instead of behind it (C3 +3.5 %, C4 +3 %, C5 +2.5 % over 8 blocks,
    // profiles/r03_d_blocks_per_cu_and_tile.txt) — unless the whole scene fits the small tile anyway (C2).
    const bool wholeSceneInSmallTile = (size_t)sc.nNodes * (size_t)c->asmNodeStride + (size_t)sc.nTriRecs * 48 <= 8192;
    const bool shareSlots = sharedGpu && !wholeSceneInSmallTile;
    const int maxBlocks = std::max(1, (c->extendMaxBlocksPerCU > 0 ? std::min(c->extendMaxBlocksPerCU, 8) : (shareSlots ? 6 : 8)) / BPW);
    const size_t tileWanted = c->extendCacheSet ? (size_t)c->extendCacheBytes : (shareSlots ? 16384 : 8192) * (size_t)BPW;
    size_t cb = std::min<size_t>(tileWanted, ldsPerCU - fixed);
    {   // the node tile gives way to residency, as in launchExtendPersist
        const int want = maxBlocks;
        const size_t perBlock = ldsPerCU / (size_t)want;
        if (fixed + cb + 16 > perBlock && perBlock > fixed + 16 + 2048) cb = std::min(cb, (perBlock - fixed - 16) & ~(size_t)15);
    }
    EpAsmArgs a{};
    const size_t NS = (size_t)c->asmNodeStride;
    a.ldsNodes = (int)std::min<size_t>((size_t)sc.nNodes, cb / NS);
    a.ldsTris = (a.ldsNodes == sc.nNodes) ? (int)std::min<size_t>((size_t)sc.nTriRecs, (cb - (size_t)a.ldsNodes * NS) / 48) : 0;
    size_t lds = (size_t)a.ldsNodes * NS + (size_t)a.ldsTris * 48 + fixed;
    lds = (lds + 15) & ~(size_t)15;
    int perCU = std::max(1, std::min((int)(ldsPerCU / lds), maxBlocks));
    int grid = cus * perCU;
    grid = std::max(1, std::min(grid, ((int)pr.launched + TPB - 1) / TPB));
    const bool fastRcp = c->streamFast && !c->debugExactExtend;
    const int variant = (c->stackMode == 2 ? (fastRcp ? 5 : 4) : (c->stackMode == 1 ? 1 : 0) + (fastRcp ? 2 : 0)) + (TPB == 1024 ? 6 : TPB == 512 ? 12 : 0);
    if (loadAsmKernel(c, variant)) { c->asmError = "hand-written intersect kernel: " + g_err; return false; }      // loud: pump() fails, no silent fallback
    a.nodes80 = c->dNodes80; a.tris = c->dTris; a.roots = c->dRoots; a.G0 = pr.st.G0; a.G1 = pr.st.G1; a.H = pr.st.H;
    a.queue = c->dQueue[pr.iter & 1]; a.ctl = c->dCtl;
    a.ellip = c->dEllip; a.numEllip = sc.numEllip; a.nodeStride = c->asmNodeStride; a.groupShift = c->asmGroupShift; a.stackDepth = c->stackDepth; a.HX = c->ellipMaps ? (void*)pr.st.HX : nullptr;
    a.numObj = sc.numObj; a.iter = pr.iter; a.nSlots = (int)pr.launched; a.refillMin = c->refillMin; a.keepEighths = c->innerKeepEighths; a.noneMin = c->noneMin;
    // main loop: the fused trip with fetch-at-decision, unless the whole scene sits in the LDS tile — then no fetch is worth hiding and the
    // phase-voting loop's fewer instructions per ray win (C2: 3.4 against 3.1 Gsamples/s, profiles/r03_c_*)
    const bool allInLds = a.ldsNodes == sc.nNodes && a.ldsTris == sc.nTriRecs;
    a.mode = c->asmLoop >= 0 ? (unsigned)c->asmLoop : (allInLds ? 0u : 1u);
    if (a.mode && !c->noneMinSet) a.noneMin = 2;      // the fused loop serves lanes that wait for their next BVH sooner (C3 +1.8 %, C5 +1 %, C4 / one stream +-0: profiles/r03_c_main_loops.txt (8))
    a.nWaves = (unsigned)grid * (unsigned)(TPB / 64);
#ifdef PT_ASM_DEBUG                  // developer builds only (scripts/build_variant.py -DPT_ASM_DEBUG): the per-wave debug records of the assembly
    {
        if (!c->dAsmDbg) { if (hipMalloc(&c->dAsmDbg, 8192 * 64) != hipSuccess) return false; }
        hipMemsetAsync(c->dAsmDbg, 0xff, 8192 * 64, pr.stream);
        a.dbg = c->dAsmDbg;
    }
#endif
    {   // x / nWaves == mulhi(x, divM) >> divS for x < 2^31 (nWaves >= 4)
        unsigned d = a.nWaves; int l = 0;
        while ((1ull << l) < d) l++;
        a.divM = (unsigned)(((1ull << (31 + l)) + d - 1) / d); a.divS = (unsigned)(l - 1);
    }
    size_t asz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    const hipError_t e = hipModuleLaunchKernel(c->asmFn[variant], (unsigned)grid, 1, 1, (unsigned)TPB, 1, 1, (unsigned)lds, pr.stream, nullptr, extra);
    if (e != hipSuccess) { c->asmError = std::string("hipModuleLaunchKernel(pt_extend_asm): ") + hipGetErrorString(e); return false; }
    c->asmLaunches++;
    return true;
}

void launchExtendPersist(pt_ctx* c, const PoolRun& pr) {
    if (c->extendMode == 2 && launchExtendAsm(c, pr)) return;
    const int launched = (int)pr.launched;
    DevScene sc = c->sc;
    int tpb = c->extendTpb;
    // LDS per block: [node tile][triangle tile][root-box distances][traversal stacks]; the tile takes what the fixed parts leave
    size_t fixed = (size_t)std::min(sc.numObj, 8) * tpb * 4 + (size_t)c->stackDepth * tpb * (c->stackMode == 2 ? 4 : 2) + 64 + 32 * 8 + (size_t)tpb * 4;     // (+ the per-lane slot numbers)     // + the LDS copies of up to 8 object roots
    size_t avail = fixed < 160 * 1024 ? 160 * 1024 - fixed : 0;
    size_t cb = std::min<size_t>((size_t)c->extendCacheBytes, avail);
    {   // a smaller node tile (down to 2 KB; 6 KB under blocks of 512 threads and more) if that lets every wave slot of the CU be used:
        // occupancy is worth more to this kernel than the last kilobytes of tile (8 instead of 6 waves per SIMD +9 %; tiles of 4, 8 and 16 KB
        // measure the same, profiles/r02_y_*)
        const int want = std::min(c->extendMaxBlocksPerCU > 0 ? c->extendMaxBlocksPerCU : 2048 / tpb, 2048 / tpb);
        const size_t perBlock = (size_t)160 * 1024 / (size_t)std::max(want, 1);
        const size_t minTile = tpb >= 512 ? 6 * 1024 : 2 * 1024;
        if (fixed + cb + 16 > perBlock && perBlock > fixed + 16 + minTile) cb = std::min(cb, (perBlock - fixed - 16) & ~(size_t)63);
    }
    sc.ldsNodes = (int)std::min<size_t>((size_t)sc.nNodes, cb / 64);
    sc.ldsTris = (sc.ldsNodes == sc.nNodes) ? (int)std::min<size_t>((size_t)sc.nTriRecs, (cb - (size_t)sc.ldsNodes * 64) / 48) : 0;
    size_t lds = (size_t)sc.ldsNodes * 64 + (size_t)sc.ldsTris * 48 + fixed;
    lds = (lds + 15) & ~(size_t)15;
    // Blocks per CU of the grid = what is resident at once (LDS per block; 2048 threads per CU), capped by pt_set_option 8 (default: no cap).
    // All waves of the grid start within 1 µs of each other (per-wave stamps of a -DPT_WAVE_STAMPS build, scripts/wave_ends.py).
    // A grid LARGER than what is resident queues blocks behind the resident ones and is slower (with 512-thread blocks: 5-16 blocks per
    // CU on C3; C4 and C5, whose deeper traversal stacks then left room for 3 blocks only, lost 5 % and 13 % with 4).
    int perCU = std::max(1, std::min((int)(160 * 1024 / std::max<size_t>(lds, 1)), 2048 / tpb));
    if (c->extendMaxBlocksPerCU > 0) perCU = std::min(perCU, c->extendMaxBlocksPerCU);
    int grid = c->numCUs * perCU;
    int maxUseful = (launched + tpb - 1) / tpb;                  // never more blocks than 1 lane per ray
    grid = std::max(1, std::min(grid, maxUseful));
#define EP(COUNT, T, TPB) launchEP<COUNT, T, TPB>(c, pr, sc, lds, grid)
#define EP_T(COUNT, T) do { if (tpb == 64) EP(COUNT, T, 64); else if (tpb == 128) EP(COUNT, T, 128); else if (tpb == 256) EP(COUNT, T, 256); else if (tpb == 512) EP(COUNT, T, 512); else EP(COUNT, T, 1024); } while (0)
    if (c->countStats) { if (c->stackMode == 0) EP_T(true, short); else if (c->stackMode == 1) EP_T(true, Packed18); else EP_T(true, int); }
    else { if (c->stackMode == 0) EP_T(false, short); else if (c->stackMode == 1) EP_T(false, Packed18); else EP_T(false, int); }
#undef EP
#undef EP_T
}

// ------------------------------------------------------------------------------------------------ frame-stream scheduler
// A batch is SUBMITTED (its jobs are appended to the running stream, or a new stream starts) and later RETIRED (all its
// pixel-frame jobs done -> k_accumulate adds its frames to the FRAME image, in u_frameCount order).  Between the two the host
// only PUMPS: it launches iterations (intersect + shade) in groups and polls the scheduler words after each group.
//   pt_render_batch        = submit + pump until the batch is retired
//   pt_render_batch_async  = submit + pump until most of its jobs have been handed out; the rest, and the jobs still in flight,
//                            are finished underneath the next batch (or by pt_finish_image / any synchronous entry point)

Batch streamBatch(const pt_ctx* c) {
    Batch b;
    b.W = c->W; b.H = c->H; b.nLocal = c->nLocal; b.nSlots = c->nSlotsImg; b.shardCount = c->shardCount;
    b.ringFrames = (unsigned)c->ringFrames; b.seeds = c->dSeeds; b.pixList = c->dPixList; b.pixXY = c->dPixXY; b.colbuf = c->dColbuf;
    if (c->nLocal >= 2) {                                     // ceil(2^(31+l)/d), exact for every job < 2^31
        unsigned d = (unsigned)c->nLocal; int l = 0;
        while ((1ull << l) < d) l++;
        unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;
        b.divM = (unsigned)m; b.divS = (unsigned)(l - 1);
    } else { b.divM = 0; b.divS = 0; }
    return b;
}

int retireFront(pt_ctx* c) {
    const pt_ctx::Entry e = c->pending.front();
    c->pending.pop_front();
    hipStream_t s = c->stream;
    Batch b = streamBatch(c);
    int gridA = (c->nSlotsImg + BLOCK - 1) / BLOCK;
    TIMED_LAUNCH(3, hipLaunchKernelGGL(k_accumulate, dim3(gridA), dim3(BLOCK), 0, s, b, c->dFc, c->dImage[e.image], e.f0, e.nFrames, e.firstFrame));
    HIP_TRY(hipGetLastError());
    return 0;
}

// The streams of destroyed contexts are kept for later contexts of the process instead of being destroyed.  hipStreamDestroy right behind a long asynchronous run —
// hundreds of commands retired moments ago, the stream synchronised, the device synchronised — is where the HIP runtime (7.0.2 as bundled with torch) went on to
// decrement a counter inside the freed 920-byte stream object: a write after free that took the host heap with it about once in a thousand call sequences (glibc
// aborts, a std::bad_variant_access out of libamdhip64).  Found with a checking allocator (tools/canary_malloc.cpp), bisected to the scheduler that no longer
// synchronises the stream at every look, cured by never destroying a stream: 2500 sequences clean (profiles/r06_f_runtime_write_after_free.txt).  Streams are few
// and small.  key: the device, or (device + 1) * 100 + e (+ 50) for the CU-masked pair of the partition experiment.
std::mutex g_streamPoolLock;
std::vector<std::pair<int, hipStream_t>> g_streamPool;
bool pooledStream(int key, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_streamPoolLock);
    for (size_t k = 0; k < g_streamPool.size(); k++)
        if (g_streamPool[k].first == key) { *out = g_streamPool[k].second; g_streamPool.erase(g_streamPool.begin() + (long)k); return true; }
    return false;
}
int takeStream(int device, hipStream_t* out) {
    if (pooledStream(device, out)) return 0;
    HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return 0;
}
void giveStream(int device, hipStream_t s) { std::lock_guard<std::mutex> lk(g_streamPoolLock); g_streamPool.emplace_back(device, s); }

// CU-masked streams of the spatial partition.  Bit k of a mask is CU k in the driver's order; whether consecutive bits walk the CUs of one XCD or the XCDs round-robin,
// the pattern ((k / 8) + (k % 8)) % 8 < e puts 4 e of every XCD's 32 CUs on the intersect side.
int ensurePartition(pt_ctx* c) {
    if (c->cuPartition == c->cuPartitionBuilt) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->sExt) { hipStreamSynchronize(c->sExt); giveStream((c->device + 1) * 100 + c->cuPartitionBuilt, c->sExt); c->sExt = nullptr; }
    if (c->sShade) { hipStreamSynchronize(c->sShade); giveStream((c->device + 1) * 100 + 50 + c->cuPartitionBuilt, c->sShade); c->sShade = nullptr; }
    c->cuPartitionBuilt = 0;
    if (c->cuPartition > 0) {
        const int words = (c->numCUs + 31) / 32;
        std::vector<uint32_t> mE((size_t)words, 0u), mS((size_t)words, 0u);
        for (int k = 0; k < c->numCUs; k++) {
            const bool ext = ((k / 8) + (k % 8)) % 8 < c->cuPartition;
            (ext ? mE : mS)[(size_t)k / 32] |= 1u << (k % 32);
        }
        if (!pooledStream((c->device + 1) * 100 + c->cuPartition, &c->sExt)) HIP_TRY(hipExtStreamCreateWithCUMask(&c->sExt, (uint32_t)words, mE.data()));
        if (!pooledStream((c->device + 1) * 100 + 50 + c->cuPartition, &c->sShade)) HIP_TRY(hipExtStreamCreateWithCUMask(&c->sShade, (uint32_t)words, mS.data()));
        if (!c->evExt) { HIP_TRY(hipEventCreateWithFlags(&c->evExt, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->evShade, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->evHost, hipEventDisableTiming)); }
        c->cuPartitionBuilt = c->cuPartition;
    }
    return 0;
}

enum PumpUntil { PUMP_IDLE, PUMP_ISSUED, PUMP_IMAGE, PUMP_RING };
// PUMP_IDLE: every batch retired.  PUMP_ISSUED: the jobs not yet handed out fit into roughly one more group of iterations
// (never waits for the pool to run dry).  PUMP_IMAGE: no unretired batch targets image `arg`.  PUMP_RING: at most `arg` ring
// rows are still owned by unretired batches.
// The oldest group in flight: its snapshot of Control, once its stamp has arrived (wait = false: only if it already has).  1 = looked at, 0 = not ready yet, < 0 = PT_ERR_*.
int processOldestGroup(pt_ctx* c, bool wait, bool discard) {
    pt_ctx::Group& g = c->grp[(c->grpHead + 2 - c->grpCount) % 2];
    // has the group's stamp arrived?  (k_snapshot writes it behind a system-scope fence after the snapshot; pinned coherent memory needs no synchronisation to be read)
    auto landed = [&]() { return *g.stamp == g.seq; };
    if (!landed()) {
        if (!wait) return 0;
        for (int spin = 0; spin < 4000 && !landed(); spin++) __builtin_ia32_pause();
        for (uint64_t n = 0; !landed(); n++) {
            std::this_thread::sleep_for(std::chrono::microseconds(n < 100 ? 20 : 100));
            if ((n & 1023) == 1023) {                              // every ~0.1 s: is the stream still working?  an idle stream without the stamp is a lost launch, an error a fault
                const hipError_t q = hipStreamQuery(c->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(PT_ERR_HIP, std::string("the wavefront stream failed: ") + hipGetErrorString(q));
                if (q == hipSuccess && !landed()) return fail(PT_ERR_HIP, "the wavefront stream is idle but a group's snapshot never arrived (internal error)");
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    c->grpCount--;
    if (g.nScan) c->scanInFlight = false;
    c->inflightPredicted = std::max<int64_t>(0, c->inflightPredicted - g.predicted);
    if (discard) return 1;
    const Control& h = *g.h;
    c->lastDelta = h.nextJob >= c->lastNextJob ? h.nextJob - c->lastNextJob : 0; c->lastCheck = g.check;
    c->lastNextJob = h.nextJob;
    bool allDead = false;
    if (g.epoch == c->submitEpoch) {                               // nothing was submitted since the group was launched: its view of the tail is the stream's
        if (h.exhausted[(g.iterEnd + 3) & 3]) {                    // the iteration after the group reads the queue: its count is exact (and only falls from there)
            c->draining = true;
            c->launched = h.qCount[32 * (g.iterEnd & 1)];
            allDead = c->launched == 0;
        } else if (h.nextJob >= c->streamJobs) {
            c->draining = true;                                    // jobs just ran out; the queue starts within two iterations
        }
    }
    int rc;
    if (allDead) { while (!c->pending.empty()) if ((rc = retireFront(c))) return rc; }
    else if (g.nScan && !c->pending.empty() && c->pending.front().f0 == g.scanF0) {
        for (int k = 0; k < g.nScan && !c->pending.empty() && !h.busy[k]; k++) if ((rc = retireFront(c))) return rc;      // in u_frameCount order, oldest first
    }
    return 1;
}
int drainGroups(pt_ctx* c, bool discard) {
    while (c->grpCount > 0) { const int r = processOldestGroup(c, true, discard); if (r < 0) return r; }
    return 0;
}

// PUMP_IDLE: every batch retired.  PUMP_ISSUED: the jobs not yet handed out fit into roughly one more group of iterations
// (never waits for the pool to run dry, and never for the group it starts).  PUMP_IMAGE: no unretired batch targets image `arg`.  PUMP_RING: at most `arg` ring
// rows are still owned by unretired batches.
int pump(pt_ctx* c, PumpUntil until, int arg) {
    if (c->pending.empty()) return drainGroups(c, true);          // (groups launched before the last batch retired ran over a dead pool: nothing to learn from them)
    hipStream_t s = c->stream;
    { int rc = ensurePartition(c); if (rc) return rc; }
    const bool part = c->sExt != nullptr && s == c->ownStream;
    hipStream_t sx = part ? c->sExt : s, ss = part ? c->sShade : s;
    // the Parameters block the running stream was started with: a later pt_set_buffer(PT_BIND_PARAMS) only takes effect with the
    // next stream (submitBatch finishes this one first), so the remaining iterations keep their kernel variants and bounds
    const float* P = c->streamIn.params;
    const bool direct = P[9] != 1.0f;
    const bool fastNow = c->streamFast;                           // the contract the running stream was started with
    const Batch b = streamBatch(c);
    const int N = c->poolActive;
    size_t ldsBytes = (size_t)c->sc.ldsNodes * 64 + (size_t)c->sc.ldsTris * 48 + (size_t)c->stackDepth * BLOCK * 4;
    auto satisfied = [&]() {
        if (c->pending.empty()) return true;
        switch (until) {
            case PUMP_IDLE: return false;
            case PUMP_ISSUED: return c->draining || (int64_t)c->streamJobs - (int64_t)c->lastNextJob - c->inflightPredicted <= std::max<int64_t>((int64_t)c->lastDelta, (int64_t)arg);
            case PUMP_IMAGE: for (const auto& e : c->pending) if (e.image == arg) return false; return true;
            case PUMP_RING: return (int)(c->streamFrames - c->pending.front().f0) <= arg;
        }
        return true;
    };
    // every job retires within SAMPLE_RES * ceil(MAX_BOUNCES) iterations of being started, and a slot runs at most
    // ceil(jobs / slots) jobs back to back: a pump that exceeds this bound (x2) is a scheduler bug, not work
    const uint64_t outstanding = (uint64_t)c->streamJobs - std::min<uint64_t>(c->lastNextJob, c->streamJobs) + (uint64_t)N;
    const uint64_t maxIters = 2 * ((outstanding + N - 1) / N + 1) * (uint64_t)(std::ceil(P[4]) * std::ceil(P[5]) + 1) + 64 + 48;
    uint64_t iters = 0;
    // One group: its iterations, a scan of the oldest batches once they have been handed out completely, the snapshot of Control with its stamp.
    // The device runs the schedule by itself: slots pull jobs while there are any; from the iteration after the first empty
    // pull on, every shading launch packs the surviving slots into a dense queue for the next iteration (Control::exhausted).
    // The host only looks: while jobs remain the end is at least one whole job (>= SAMPLE_RES iterations) away, so a group is
    // 24 iterations, in the tail 8; each look shrinks the launch grids to the live count.
    auto launchGroup = [&](bool kick) -> int {
        int CHECK = c->draining ? 8 : 24;
        if (kick && c->lastDelta == 0) CHECK = 4;                  // the first looks of a stream fed in small submissions come early: the pool grows with the backlog they report
        const int64_t perIter = std::max<int64_t>(1, (int64_t)c->lastDelta / std::max(1, c->lastCheck));      // jobs handed out per iteration at the last look
        if (until == PUMP_ISSUED && c->lastDelta > 0) {           // approach the end of the job supply without running into it
            const int64_t backlog = std::max<int64_t>(0, (int64_t)c->streamJobs - (int64_t)c->lastNextJob - c->inflightPredicted);      // as of the last look, less what the groups in flight take
            const int64_t left = backlog - (int64_t)c->lastDelta / 2 - (int64_t)arg;
            if (left > 0) CHECK = (int)std::max<int64_t>(1, std::min<int64_t>(CHECK, left / perIter));
            else if (kick) CHECK = (int)std::max<int64_t>(1, std::min<int64_t>(CHECK, std::max<int64_t>(backlog, (int64_t)c->lastSubmitJobs) / perIter - 2));      // (two iterations' worth stay behind: later submissions sit BEHIND this group in the stream, and a pull that comes back empty sends the pool into its tail)
        }
        if (part) { HIP_TRY(hipEventRecord(c->evHost, s)); HIP_TRY(hipStreamWaitEvent(sx, c->evHost, 0)); }      // what `s` holds (submission, revive, accumulate) comes first
        for (int k = 0; k < CHECK; k++) {
            PoolRun pr; pr.stream = sx; pr.st = c->st; pr.launched = c->launched; pr.iter = c->iter;
            const int grid = std::max(1, (int)((pr.launched + BLOCK - 1) / BLOCK));
            if (c->extendMode == 0) {
                if (c->countStats) TIMED_LAUNCH_ON(sx, 0, hipLaunchKernelGGL(k_extend<true>, dim3(grid), dim3(BLOCK), ldsBytes, sx, c->sc, pr.st, c->dQueue[pr.iter & 1], pr.iter, (int)pr.launched, c->dCtl));
                else TIMED_LAUNCH_ON(sx, 0, hipLaunchKernelGGL(k_extend<false>, dim3(grid), dim3(BLOCK), ldsBytes, sx, c->sc, pr.st, c->dQueue[pr.iter & 1], pr.iter, (int)pr.launched, c->dCtl));
            } else {
                TIMED_LAUNCH_ON(sx, 0, launchExtendPersist(c, pr));
            }
            if (part) { HIP_TRY(hipEventRecord(c->evExt, sx)); HIP_TRY(hipStreamWaitEvent(ss, c->evExt, 0)); }
#define SHADE_ARGS dim3(std::max(1, (int)((pr.launched + SHADE_BLOCK - 1) / SHADE_BLOCK))), dim3(SHADE_BLOCK), 0, ss, c->sc, b, c->dFc, pr.st, c->dQueue[pr.iter & 1], c->dQueue[(pr.iter + 1) & 1], pr.iter, (int)pr.launched, c->dCtl
            // kernel variant: transmissive materials present / statistics on / RAYTRACING == 0 / texture-mapped materials present
            // (T: index-stack encoding of the path state — 0 no transmissive material, 3 / 8 bits per slot)
#define SHADE_V(T, S, D, X) TIMED_LAUNCH_ON(ss, 1, hipLaunchKernelGGL((k_shade<T, S, D, X>), SHADE_ARGS))
#define SHADE_F(T, X) TIMED_LAUNCH_ON(ss, 1, hipLaunchKernelGGL((k_shade<T, false, false, X, true>), SHADE_ARGS))
            // (the relaxed numeric contract, pt_set_option 16: path tracing without statistics only; everything else keeps the exact kernels)
#define SHADE_S(T, D, X) do { if (c->countStats) SHADE_V(T, true, D, X); else if (fastNow && !(D)) SHADE_F(T, X); else SHADE_V(T, false, D, X); } while (0)
#define SHADE_X(T, D) do { if (c->anyMaps) SHADE_S(T, D, true); else SHADE_S(T, D, false); } while (0)
            if (direct) SHADE_X(0, true);
            else if (c->niBits == 3) SHADE_X(3, false);
            else if (c->niBits == 8) SHADE_X(8, false);
            else if (c->niBits == 32) SHADE_X(32, false);
            else SHADE_X(0, false);
            if (part) { HIP_TRY(hipEventRecord(c->evShade, ss)); HIP_TRY(hipStreamWaitEvent(sx, c->evShade, 0)); }
            c->iter = (c->iter + 1) & 0x3fffffff;
            iters++;
        }
        if (part) HIP_TRY(hipStreamWaitEvent(s, c->evShade, 0));
        HIP_TRY(hipGetLastError());                                // a failed launch surfaces here, not as "did not drain"
        if (!c->asmError.empty()) { const std::string m = c->asmError; c->asmError.clear(); return fail(PT_ERR_HIP, m); }
        pt_ctx::Group& g = c->grp[c->grpHead];
        g.check = CHECK; g.iterEnd = c->iter; g.epoch = c->submitEpoch; g.nScan = 0;
        g.predicted = c->lastDelta > 0 ? (int64_t)CHECK * perIter : 0; c->inflightPredicted += g.predicted;
        // have the oldest batches been handed out completely (as of the last look)?  then see which of them are still in flight (one scan in flight at a time)
        if (!c->scanInFlight) {
            ScanEnds ends{};
            for (const auto& e : c->pending) {
                if (ends.n == 8 || c->lastNextJob < e.jobEnd) break;
                ends.f[ends.n++] = e.f0 + (unsigned)e.nFrames;
            }
            if (ends.n) {
                HIP_TRY(hipMemsetAsync(c->dCtl->busy, 0, sizeof(c->dCtl->busy), s));
                hipLaunchKernelGGL(k_scan_inflight, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, c->st, N, ends, c->dCtl);
                g.nScan = ends.n; g.scanF0 = c->pending.front().f0; c->scanInFlight = true;
            }
        }
        if (++c->groupSeq == 0) c->groupSeq = 1;                   // (0 = "nothing has arrived")
        g.seq = c->groupSeq; *g.stamp = 0;
        hipLaunchKernelGGL(k_snapshot, dim3(1), dim3(64), 0, s, c->dCtl, g.h, g.stamp, g.seq);
        HIP_TRY(hipGetLastError());
        c->grpHead = (c->grpHead + 1) % 2; c->grpCount++;
        return 0;
    };
    int rc;
    bool kick = until == PUMP_ISSUED;                             // a submission always gets the GPU going: about as many iterations as consume what it added
    while (c->grpCount > 0 && (rc = processOldestGroup(c, false, false)) != 0) if (rc < 0) return rc;      // whatever has finished since the last call
    for (;;) {
        const bool want = kick || !satisfied();
        if (!want) break;
        if (iters > maxIters) return fail(PT_ERR_HIP, "wavefront scheduler did not drain (internal error)");
        const int room = c->draining ? 1 : 2;                     // the tail is run look by look: every look shrinks the grids
        if (c->grpCount < room) { if ((rc = launchGroup(kick))) return rc; kick = false; continue; }
        if (kick) { kick = false; continue; }                     // two groups are on their way already: the submission rides behind them
        if ((rc = processOldestGroup(c, true, false)) < 0) return rc;
    }
    c->hostCnt[PT_CNT_ITERATIONS] += iters;
    c->hostCnt[PT_CNT_EXTEND_LAUNCHES] += iters;
    if (until != PUMP_ISSUED) return drainGroups(c, c->pending.empty());      // synchronous callers leave nothing behind them
    return 0;
}

int flushStream(pt_ctx* c) {
    if (c->pending.empty()) return 0;
    HIP_TRY(hipSetDevice(c->device));
    return pump(c, PUMP_IDLE, 0);
}

int submitBatch(pt_ctx* c, int firstFrame, int nFrames, const int32_t* seeds, bool async) {
    if (nFrames < 1) return fail(PT_ERR_ARG, "n_frames must be >= 1");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc;
    // parameter checks (scope: SURVEY.md §2)
    if (c->params.size() < 12) return fail(PT_ERR_ARG, "Parameters (binding 4) not set");
    const float* P = c->params.data();
    const bool direct = P[9] != 1.0f;                            // RAYTRACING == 0: directDiffuse (frag.glsl:655-681, :911-912)
    if (P[10] != 0.0f) {                                          // DEBUG: no paths at all, one small kernel (frag.glsl:916-918)
        if ((int)P[2] != c->W || (int)(P[2] * P[3]) != c->H) return fail(PT_ERR_ARG, "Parameters.resolution / screenHratio do not match the FRAME image size given to pt_create");
        if ((rc = flushStream(c))) return rc;
        if (c->sceneDirty && (rc = buildScene(c))) return rc;
        if (c->stackDepth > 64) return fail(PT_ERR_SCENE, "DEBUG heat-map: BVH deeper than the 64-entry traversal stack");
        FrameIn fin;
        std::memcpy(fin.params, P, 48); std::memcpy(fin.origin, c->origin.data(), 12); std::memcpy(fin.rotation, c->rotation.data(), 12); std::memcpy(fin.mouse, c->mouse.data(), 12);
        *c->hFrameIn = fin;
        HIP_TRY(hipMemcpyAsync(c->dFrameIn, c->hFrameIn, sizeof(FrameIn), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_frame_setup, dim3(1), dim3(64), 0, s, c->sc, c->dFrameIn, c->dFc, c->dEllip);
        std::memset(&c->streamIn, 0xff, sizeof(FrameIn));                          // the frame constants on the device are no stream's any more
        const Batch b = streamBatch(c);
        DevScene dsc = c->sc; dsc.ldsNodes = 0; dsc.ldsTris = 0;                    // no LDS tile in this kernel
        hipLaunchKernelGGL(k_debug_heatmap, dim3((c->nLocal + 63) / 64), dim3(64), 0, s, dsc, b, c->dFc, c->dImage[c->curImage], firstFrame, nFrames);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if ((int)P[2] != c->W || (int)(P[2] * P[3]) != c->H) return fail(PT_ERR_ARG, "Parameters.resolution / screenHratio do not match the FRAME image size given to pt_create");
    // the loop bounds are floats in the shader (frag.glsl:820, :898); the slot's counters have 12 bits each
    if (!(P[4] >= 1.0f) || P[4] > 2047.0f) return fail(PT_ERR_ARG, "SAMPLE_RES must be in [1,2047]");
    if (!(P[5] > 0.0f) || P[5] > 4095.0f) return fail(PT_ERR_ARG, "MAX_BOUNCES must be in (0,4095]");
    size_t nJobs64 = (size_t)c->nLocal * (size_t)nFrames;
    if (nJobs64 >= (1ull << 31)) return fail(PT_ERR_ARG, "batch too large: pixels * frames must stay below 2^31 (split the batch)");
    FrameIn fin;
    std::memcpy(fin.params, P, 48); std::memcpy(fin.origin, c->origin.data(), 12); std::memcpy(fin.rotation, c->rotation.data(), 12); std::memcpy(fin.mouse, c->mouse.data(), 12);
    // the running stream can take this batch if nothing the kernels were launched with changes
    // overlapped: room for the batches of as many images as can be pending, and for callers that submit frame by frame to run
    // ahead (at least 64 rows while they stay below 8 GB)
    int wantRing = nFrames;
    if (async) {
        const size_t rowBytes = (size_t)c->nSlotsImg * 16;
        wantRing = std::max(pt_ctx::IMAGES * nFrames, (int)std::min<size_t>(64, std::max<size_t>(1, ((size_t)8 << 30) / rowBytes)));
    }
    bool join = !c->pending.empty() && !c->sceneDirty && std::memcmp(&fin, &c->streamIn, sizeof(FrameIn)) == 0 && c->ringFrames >= wantRing && c->streamFast == c->fastContract &&
                (uint64_t)c->streamJobs + nJobs64 < (1ull << 31);
    if (!join && (rc = flushStream(c))) return rc;
    if (!join) {                                                  // ---- a new stream
        if (c->sceneDirty && (rc = buildScene(c))) return rc;
        if (direct && c->anySubsurface) {
            if (c->sc.numObj > FL_PROBE_OBJ_MAX + 1) return fail(PT_ERR_UNSUPPORTED, "directDiffuse with subsurface materials supports at most 65536 objects (BVHs)");
            if (c->ambiguousTriObj) return fail(PT_ERR_SCENE, "directDiffuse with subsurface materials needs every triangle to belong to one BVH (hit.parentID, frag.glsl:573)");
        }
        if (c->poolSlots > 0) c->poolActive = c->poolSlots;
        else {                                                    // automatic pool (measured on C3, profiles/): 5/8 of the batch up to 2^23 when batches overlap
            // (a batch that drains: one slot per job up to 2^22 — a frame at a time, the reference's own loop, takes 16.0 instead of 19.2 ms per
            //  1080p frame with 2 M instead of 1 M slots, profiles/r02_m_frame_at_a_time_loop.txt: every job then runs from the first iteration)
            size_t want = std::min<size_t>(std::max<size_t>(async ? nJobs64 * 5 / 8 : nJobs64, (size_t)1 << 20), (size_t)1 << (async ? 23 : 22));
            c->poolActive = (int)((std::min<size_t>(want, std::max<size_t>(nJobs64, BLOCK)) + BLOCK - 1) / BLOCK * BLOCK);
        }
        if ((rc = ensurePool(c, (async && c->poolSlots == 0) ? (1 << 23) : 0))) return rc;
        if (c->ringFrames < wantRing) {
            HIP_TRY(hipStreamSynchronize(s));
            if (c->dColbuf) HIP_TRY(hipFree(c->dColbuf));
            if (c->dSeeds) HIP_TRY(hipFree(c->dSeeds));
            if (c->hSeeds) HIP_TRY(hipHostFree(c->hSeeds));
            c->dColbuf = nullptr; c->dSeeds = nullptr; c->hSeeds = nullptr; c->ringFrames = 0;
            HIP_TRY(hipMalloc((void**)&c->dColbuf, (size_t)wantRing * (size_t)c->nSlotsImg * 16));
            HIP_TRY(hipMalloc((void**)&c->dSeeds, (size_t)wantRing * 4));
            HIP_TRY(hipHostMalloc((void**)&c->hSeeds, (size_t)wantRing * 4, hipHostMallocDefault));
            c->ringFrames = wantRing;
        }
        c->streamIn = fin; *c->hFrameIn = fin; c->streamFast = c->fastContract;
        HIP_TRY(hipMemcpyAsync(c->dFrameIn, c->hFrameIn, sizeof(FrameIn), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_frame_setup, dim3(1), dim3(64), 0, s, c->sc, c->dFrameIn, c->dFc, c->dEllip);
        hipLaunchKernelGGL(k_init_control, dim3(1), dim3(1), 0, s, c->dCtl);
        HIP_TRY(hipMemsetAsync(c->st.G1, 0, (size_t)c->poolActive * 16, s));       // every slot dead
        if ((rc = drainGroups(c, true))) return rc;
        c->inflightPredicted = 0;
        c->streamFrames = 0; c->streamJobs = 0; c->lastNextJob = 0; c->lastDelta = 0; c->lastCheck = 24; c->iter = 0;
    } else if ((int)(c->streamFrames - c->pending.front().f0) + nFrames > c->ringFrames) {
        if ((rc = pump(c, PUMP_RING, c->ringFrames - nFrames))) return rc;        // wait for ring rows
        if (c->pending.empty()) return submitBatch(c, firstFrame, nFrames, seeds, async);   // the stream ended meanwhile: start over
    }
    // a stream fed in small batches (the reference draws ONE frame per call) started with a small pool: let it grow with the backlog
    bool grown = false;
    if (join && async && c->poolSlots == 0) {
        const uint64_t outstanding = (uint64_t)std::max<int64_t>(0, (int64_t)c->streamJobs - (int64_t)std::min<uint64_t>(c->lastNextJob, c->streamJobs) - c->inflightPredicted) + nJobs64;
        size_t target = std::min<size_t>(std::max<size_t>(outstanding * 5 / 8, (size_t)1 << 20), (size_t)1 << 23);
        // an image is cheapest to finish two images later (pt_finish_image): keep an image's jobs worth several pool turnovers
        if (c->jobsPerImage) target = std::min<size_t>(target, std::max<size_t>((size_t)(c->jobsPerImage * 5 / 8), (size_t)1 << 20));
        target = std::min<size_t>((target + BLOCK - 1) / BLOCK * BLOCK, (size_t)c->allocSlots);
        const size_t cap = std::min<size_t>((size_t)1 << 23, (size_t)c->allocSlots);
        if (target > (size_t)c->poolActive + (size_t)c->poolActive / 4 || (target >= cap && target > (size_t)c->poolActive)) {      // (the last step to the largest pool may be a small one)
            HIP_TRY(hipMemsetAsync(c->st.G1 + c->poolActive, 0, (target - (size_t)c->poolActive) * 16, s));      // the new slots are dead
            c->poolActive = (int)target;
            grown = true;
        }
    }
    // ---- append
    const unsigned f0 = c->streamFrames;
    for (int f = 0; f < nFrames; f++) c->hSeeds[(f0 + (unsigned)f) % (unsigned)c->ringFrames] = seeds[f];
    {
        unsigned r0 = f0 % (unsigned)c->ringFrames, n0 = std::min<unsigned>((unsigned)nFrames, (unsigned)c->ringFrames - r0);
        HIP_TRY(hipMemcpyAsync(c->dSeeds + r0, c->hSeeds + r0, (size_t)n0 * 4, hipMemcpyHostToDevice, s));
        if (n0 < (unsigned)nFrames) HIP_TRY(hipMemcpyAsync(c->dSeeds, c->hSeeds, (size_t)(nFrames - n0) * 4, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_submit, dim3(1), dim3(1), 0, s, c->dCtl, (unsigned)nJobs64, join ? (grown ? 2 : 0) : 1, (unsigned)c->poolActive);
    const Batch b = streamBatch(c);
    const int N = c->poolActive;
#define REVIVE(T, F) TIMED_LAUNCH(2, hipLaunchKernelGGL((k_revive<T, F>), dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, b, c->dFc, c->st, N, c->dCtl))
    const bool fastRevive = c->streamFast && !direct && !c->countStats;
    // (directDiffuse never touches the index stack: its k_shade variant is the one without it, and so is its path state — except that the pool of a
    //  scene with transmissive materials has the groups allocated, which k_revive<niBits> initialises)
    if (c->niBits == 3) { if (fastRevive) REVIVE(3, true); else REVIVE(3, false); }
    else if (c->niBits == 8) { if (fastRevive) REVIVE(8, true); else REVIVE(8, false); }
    else if (c->niBits == 32) { if (fastRevive) REVIVE(32, true); else REVIVE(32, false); }
    else { if (fastRevive) REVIVE(0, true); else REVIVE(0, false); }
#undef REVIVE
    c->streamFrames += (unsigned)nFrames; c->streamJobs += (unsigned)nJobs64;
    c->lastSubmitJobs = nJobs64; c->jobsThisImage += nJobs64;
    pt_ctx::Entry e; e.jobEnd = c->streamJobs; e.f0 = f0; e.nFrames = nFrames; e.firstFrame = firstFrame; e.image = c->curImage;
    c->pending.push_back(e);
    c->draining = false; c->launched = (unsigned)N;              // (if the pool had run dry, k_submit dropped the tail queue)
    c->submitEpoch++;                                            // the groups in flight were launched for another tail: their view of it no longer counts
    HIP_TRY(hipGetLastError());
    // asynchronous: come back while the backlog of jobs not yet handed out is below what keeps the largest pool fed (2^23 * 8/5)
    if (async) return pump(c, PUMP_ISSUED, c->poolSlots == 0 ? 14000000 : 0);
    return pump(c, PUMP_IDLE, 0);
}

int resolveTimes(pt_ctx* c) {
    for (auto& k : c->kt) {
        for (size_t i = 0; i < k.used; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, k.ev[i].first, k.ev[i].second));
            k.ms += ms; k.launches++; k.each.push_back(ms);
        }
        k.used = 0;
    }
    return 0;
}

}  // namespace

#include "pt_multi.hpp"

// ------------------------------------------------------------------------------------------------ C ABI

int pt_set_error_(int code, const std::string& msg) { return fail(code, msg); }      // for pt_bvh.hip

// a group context hands the call to the host thread of every device context and joins them (pt_multi.hpp)
#define MULTI_ALL(c, call) do { if ((c) && (c)->multi) return multiRun(*(c)->multi, [=](pt_ctx* k) { return call; }); } while (0)
// ... for the entry points that render: not while an upload reached only some of the streams
#define MULTI_RENDER(c, call) do { if ((c) && (c)->multi) { if (!(c)->multi->staleBindings.empty()) return fail(PT_ERR_SCENE, "an earlier pt_set_buffer / pt_set_texture failed after it had reached some of the context's streams: repeat that upload before rendering"); \
                                                          return multiRun(*(c)->multi, [=](pt_ctx* k) { return call; }); } } while (0)

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }

namespace {
int initContext(pt_ctx* c, int width, int height, int shard_rank, int shard_count) {
    { const int rc = takeStream(c->device, &c->ownStream); if (rc) return rc; }
    c->stream = c->ownStream;
    shardPixels(width, height, shard_rank, shard_count, c->pixList);
    c->nLocal = (int)c->pixList.size();
    c->nSlotsImg = shard_count == 1 ? width * height : (int)shardSlots(width, height, shard_count);
    if (c->nLocal == 0) return fail(PT_ERR_ARG, "this shard owns no pixels (more shards than tiles)");
    HIP_TRY(hipMalloc((void**)&c->dPixList, (size_t)c->nLocal * 4));
    HIP_TRY(hipMemcpy(c->dPixList, c->pixList.data(), (size_t)c->nLocal * 4, hipMemcpyHostToDevice));
    {
        std::vector<unsigned> xy(c->pixList.size());
        for (size_t k = 0; k < xy.size(); k++) xy[k] = (unsigned)(c->pixList[k] % width) | ((unsigned)(c->pixList[k] / width) << 16);
        HIP_TRY(hipMalloc((void**)&c->dPixXY, xy.size() * 4));
        HIP_TRY(hipMemcpy(c->dPixXY, xy.data(), xy.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void**)&c->dImage[0], (size_t)c->nSlotsImg * 16));
    HIP_TRY(hipMemset(c->dImage[0], 0, (size_t)c->nSlotsImg * 16));
    HIP_TRY(hipMalloc((void**)&c->dFrameIn, sizeof(FrameIn)));
    HIP_TRY(hipMalloc((void**)&c->dFc, sizeof(FrameConst)));
    HIP_TRY(hipMalloc((void**)&c->dCtl, sizeof(Control)));
    HIP_TRY(hipMemset(c->dCtl, 0, sizeof(Control)));
    for (auto& g : c->grp) {
        HIP_TRY(hipHostMalloc((void**)&g.h, sizeof(Control), hipHostMallocCoherent));
        HIP_TRY(hipHostMalloc((void**)&g.stamp, 64, hipHostMallocCoherent));
        *g.stamp = 0;
    }
    HIP_TRY(hipHostMalloc((void**)&c->hFrameIn, sizeof(FrameIn), hipHostMallocDefault));
    return 0;
}
}  // namespace

int pt_create(pt_ctx** out, int device, int width, int height, int shard_rank, int shard_count) {
    if (!out || width < 1 || height < 1 || width > 65535 || height > 65535 || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return fail(PT_ERR_ARG, "pt_create: bad argument");
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev < 1) return fail(PT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= nDev) return fail(PT_ERR_NO_DEVICE, "HIP device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return fail(PT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    pt_ctx* c = new pt_ctx();
    c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->device = device; c->W = width; c->H = height; c->shardRank = shard_rank; c->shardCount = shard_count;
    c->imp = {0.0f}; c->ellip = {0.0f}; c->objidx = {0};
    c->mouse = {-1.0e6f, -1.0e6f, 0.0f};
    const int rc = initContext(c, width, height, shard_rank, shard_count);
    if (rc) { const std::string msg = g_err; pt_destroy(c); return fail(rc, msg); }      // nothing of a half-built context is left behind
    *out = c;
    return PT_OK;
}

int pt_create_multi_part(pt_ctx** out, const int* devices, int n_devices, int width, int height, int first_shard, int total_shards) {
    if (!out || !devices || n_devices < 1 || n_devices > 64 || width < 1 || height < 1 || first_shard < 0 || total_shards < n_devices || first_shard + n_devices > total_shards)
        return fail(PT_ERR_ARG, "pt_create_multi: bad argument");
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev < 1) return fail(PT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    for (int i = 0; i < n_devices; i++) if (devices[i] < 0 || devices[i] >= nDev) return fail(PT_ERR_NO_DEVICE, "pt_create_multi: HIP device index out of range");
    // the streams of one device are adjacent entries, every device carries the same number of them
    std::vector<MultiCtx::Run> runs;
    for (int i = 0; i < n_devices; i++) {
        if (!runs.empty() && runs.back().device == devices[i]) { runs.back().count++; continue; }
        for (const auto& r : runs) if (r.device == devices[i]) return fail(PT_ERR_ARG, "pt_create_multi: the entries of one device must be adjacent ({0,0,1,1}, not {0,1,0,1})");
        runs.push_back(MultiCtx::Run{devices[i], i, 1, nullptr});
    }
    for (const auto& r : runs) if (r.count != runs[0].count) return fail(PT_ERR_ARG, "pt_create_multi: every device must be listed the same number of times");
    // Test mode (tests/test_gpu_multi.py): PT_MULTI_VIRTUAL_DEVICES=k splits the streams of ONE device into k "virtual devices", so that a
    // one-GPU box runs the several-device code of the gather — a staging block per device, root and non-root arguments, block offsets,
    // the un-tiling of a gathered buffer — with the RCCL calls replaced by device copies to the offsets ncclGather writes (pt_multi.hpp).
    int virtualDevices = 0;
    if (const char* vd = getenv("PT_MULTI_VIRTUAL_DEVICES")) {
        virtualDevices = atoi(vd);
        if (virtualDevices > 1) {
            if (runs.size() != 1 || n_devices % virtualDevices) return fail(PT_ERR_ARG, "PT_MULTI_VIRTUAL_DEVICES=k needs ONE device listed a multiple of k times");
            const int per = n_devices / virtualDevices, dev = runs[0].device;
            runs.clear();
            for (int v = 0; v < virtualDevices; v++) runs.push_back(MultiCtx::Run{dev, v * per, per, nullptr});
        } else virtualDevices = 0;
    }
    pt_ctx* g = new pt_ctx();
    g->W = width; g->H = height; g->device = devices[0];
    MultiCtx* M = new MultiCtx();
    g->multi = M;
    M->n = n_devices; M->devices.assign(devices, devices + n_devices); M->runs = runs;
    M->shardBase = first_shard; M->shardTotal = total_shards;
    const char* force = getenv("PT_MULTI_FORCE_RCCL");          // tests: the RCCL call path of a one-device group on a one-GPU box
    M->useRccl = runs.size() > 1 || (force && force[0] == '1');
    M->virtualDevices = virtualDevices > 1;
    M->kids.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; i++) {
        M->workers.emplace_back(new Worker());
        Worker* w = M->workers.back().get();
        w->th = std::thread([w] { w->loop(); });
    }
    for (int i = 0; i < n_devices; i++) {
        pt_ctx** slot = &M->kids[i]; const int dev = devices[i];
        M->workers[i]->post([=] { return pt_create(slot, dev, width, height, first_shard + i, total_shards); });
    }
    int rc = 0; std::string err;
    for (int i = 0; i < n_devices; i++) { int r = M->workers[i]->wait(); if (r && !rc) { rc = r; err = "device " + std::to_string(devices[i]) + ": " + M->workers[i]->err; } }
    if (rc) { M->kids.erase(std::remove(M->kids.begin(), M->kids.end(), nullptr), M->kids.end()); multiFree(g); delete g; return fail(rc, err); }
    for (const auto& r : runs) for (int j = 0; j < r.count; j++) M->kids[r.first + j]->streamsOnDevice = virtualDevices > 1 ? n_devices : r.count;
    *out = g;
    // Streams that share a GPU overlap only on different hardware queues.  The HIP runtime deals streams to queues round-robin when
    // GPU_MAX_HW_QUEUES is set — a variable it reads ONCE, when the process initialises HIP, so it is the host's to set;
    // the library does not touch the process environment.  Without it the context works, at the speed of one stream: said here, once,
    // as a warning that pt_last_error() returns after this successful call.
    g_err.clear();
    if (n_devices > (int)runs.size() && !getenv("GPU_MAX_HW_QUEUES"))
        g_err = "warning: several streams share a GPU but GPU_MAX_HW_QUEUES is not set in the environment: the HIP runtime may put them on one hardware queue "
                "and their kernels will not overlap (set GPU_MAX_HW_QUEUES=8 before the process first uses HIP)";
    return PT_OK;
}

int pt_create_multi(pt_ctx** out, const int* devices, int n_devices, int width, int height) {
    return pt_create_multi_part(out, devices, n_devices, width, height, 0, n_devices);
}

int pt_destroy(pt_ctx* c) {
    if (!c) return PT_OK;
    if (c->multi) { multiFree(c); delete c; return PT_OK; }
    hipSetDevice(c->device);
    flushStream(c);
    hipStreamSynchronize(c->stream);
    for (hipModule_t m : c->asmModule) if (m) hipModuleUnload(m);
    void* ptrs[] = {c->dNiTable, c->st.J, c->dNodes80, c->dTexels, c->dTexTable, c->dTriObj, c->dNodes, c->dTris, c->dShade, c->dRoots, c->dEllip, c->dMats, c->dSky, c->dPixList, c->dPixXY, c->dAllMaps, c->dImage[0], c->dImage[1], c->dImage[2], c->dImage[3], c->st.G0, c->st.G1, c->st.G2,
                    c->st.G3, c->st.G4, c->st.G5, c->st.S0, c->st.H, c->st.HX, c->dQueue[0], c->dQueue[1], c->dColbuf, c->dSeeds, c->dFrameIn, c->dFc, c->dCtl, c->dDisplay};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& g : c->grp) { if (g.h) hipHostFree(g.h); if (g.stamp) hipHostFree((void*)g.stamp); }
    if (c->hFrameIn) hipHostFree(c->hFrameIn);
    if (c->hSeeds) hipHostFree(c->hSeeds);
    for (auto& k : c->kt) for (auto& e : k.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (c->sExt) { hipStreamSynchronize(c->sExt); giveStream((c->device + 1) * 100 + c->cuPartitionBuilt, c->sExt); }
    if (c->sShade) { hipStreamSynchronize(c->sShade); giveStream((c->device + 1) * 100 + 50 + c->cuPartitionBuilt, c->sShade); }
    for (hipEvent_t e : {c->evExt, c->evShade, c->evHost}) if (e) hipEventDestroy(e);
    if (c->ownStream) { hipStreamSynchronize(c->ownStream); giveStream(c->device, c->ownStream); }
    delete c;
    return PT_OK;
}

int pt_set_buffer(pt_ctx* c, int binding, const void* data, size_t bytes) {
    if (!c || (!data && bytes)) return fail(PT_ERR_ARG, "pt_set_buffer: null argument");
    if (c->multi) {                                               // the scene is replicated
        for (pt_ctx* k : c->multi->kids) {
            int rc = pt_set_buffer(k, binding, data, bytes);
            // the checks are the same for every stream, so the first one refuses a bad upload before anything changed; a failure later on
            // (out of memory) leaves the replicas different: no render until this binding has reached all of them
            if (rc) { if (k != c->multi->kids[0]) c->multi->staleBindings.insert(binding); return rc; }
        }
        c->multi->staleBindings.erase(binding);
        return PT_OK;
    }
    if (bytes % 4) return fail(PT_ERR_ARG, "pt_set_buffer: size must be a multiple of 4 bytes");
    const float* f = static_cast<const float*>(data); const int32_t* i = static_cast<const int32_t*>(data); size_t n = bytes / 4;
    switch (binding) {
        case PT_BIND_ORIGIN: if (n < 3) return fail(PT_ERR_ARG, "ORIGIN needs 3 floats"); c->origin.assign(f, f + 3); return PT_OK;      // per-frame glBufferSubData: no scene rebuild
        case PT_BIND_ROTATION: if (n < 3) return fail(PT_ERR_ARG, "ROTATION needs 3 floats"); c->rotation.assign(f, f + 3); return PT_OK;
        case PT_BIND_MOUSE: if (n < 3) return fail(PT_ERR_ARG, "MOUSE_POS needs 3 floats"); c->mouse.assign(f, f + 3); return PT_OK;
        case PT_BIND_PARAMS: if (n < 12) return fail(PT_ERR_ARG, "Parameters needs 12 floats"); c->params.assign(f, f + 12); return PT_OK;
        case PT_BIND_TRIANGLES: if (n % 40) return fail(PT_ERR_ARG, "triangle buffer must be 40 floats per triangle"); c->tris.assign(f, f + n); break;
        case PT_BIND_IMPLICITS: c->imp.assign(f, f + n); break;
        case PT_BIND_ELLIPSOIDS: c->ellip.assign(f, f + n); break;
        case PT_BIND_BVHDATA: c->bvhdata.assign(f, f + n); break;
        case PT_BIND_BVHTREE: if (n % 3) return fail(PT_ERR_ARG, "BVHtree must be 3 ints per node"); c->bvhtree.assign(i, i + n); break;
        case PT_BIND_LEAFTRIS: c->leaftris.assign(i, i + n); break;
        case PT_BIND_OBJINDICES: c->objidx.assign(i, i + n); break;
        case PT_BIND_MATERIALS: c->mtl.assign(f, f + n); break;
        default: return fail(PT_ERR_ARG, "pt_set_buffer: binding point not consumed by the render path (frag.glsl declares 0-5,7,10-15)");
    }
    c->sceneDirty = true;
    return PT_OK;
}

int pt_set_texture(pt_ctx* c, int index, int w, int h, const uint8_t* rgba8) {
    if (!c || !rgba8 || w < 1 || h < 1) return fail(PT_ERR_ARG, "pt_set_texture: bad argument");
    if (c->multi) {
        for (pt_ctx* k : c->multi->kids) {
            int rc = pt_set_texture(k, index, w, h, rgba8);
            if (rc) { if (k != c->multi->kids[0]) c->multi->staleBindings.insert(1000 + index); return rc; }
        }
        c->multi->staleBindings.erase(1000 + index);
        return PT_OK;
    }
    if (index < 0 || index > 4095) return fail(PT_ERR_ARG, "texture index out of range [0,4095]");
    if (index == 0) { c->sky.assign(rgba8, rgba8 + (size_t)w * h * 4); c->skyW = w; c->skyH = h; }
    if ((size_t)index >= c->textures.size()) c->textures.resize((size_t)index + 1);
    c->textures[index].rgba.assign(rgba8, rgba8 + (size_t)w * h * 4); c->textures[index].w = w; c->textures[index].h = h;
    c->sceneDirty = true;
    return PT_OK;
}

int pt_reset_frame(pt_ctx* c) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    MULTI_ALL(c, pt_reset_frame(k));
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = flushStream(c))) return rc;
    HIP_TRY(hipMemsetAsync(c->dImage[c->curImage], 0, (size_t)c->nSlotsImg * 16, c->stream));
    return PT_OK;
}

int pt_render(pt_ctx* c, int frame_count, int seed) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    MULTI_RENDER(c, pt_render(k, frame_count, seed));
    int32_t s = seed;
    return submitBatch(c, frame_count, 1, &s, false);
}
int pt_render_batch(pt_ctx* c, int first_frame, int n_frames, const int32_t* seeds) {
    if (!c || !seeds) return fail(PT_ERR_ARG, "pt_render_batch: null argument");
    MULTI_RENDER(c, pt_render_batch(k, first_frame, n_frames, seeds));
    return submitBatch(c, first_frame, n_frames, seeds, false);
}
int pt_render_batch_async(pt_ctx* c, int first_frame, int n_frames, const int32_t* seeds) {
    if (!c || !seeds) return fail(PT_ERR_ARG, "pt_render_batch_async: null argument");
    MULTI_RENDER(c, pt_render_batch_async(k, first_frame, n_frames, seeds));
    return submitBatch(c, first_frame, n_frames, seeds, true);
}

int pt_next_image(pt_ctx* c) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    MULTI_ALL(c, pt_next_image(k));
    HIP_TRY(hipSetDevice(c->device));
    const int next = (c->curImage + 1) % pt_ctx::IMAGES;
    if (!c->dImage[next]) HIP_TRY(hipMalloc((void**)&c->dImage[next], (size_t)c->nSlotsImg * 16));
    int rc;
    if ((rc = pump(c, PUMP_IMAGE, next))) return rc;              // nothing may still be on its way into the image taken over
    c->curImage = next;
    if (c->jobsThisImage) c->jobsPerImage = c->jobsThisImage;
    c->jobsThisImage = 0;
    HIP_TRY(hipMemsetAsync(c->dImage[next], 0, (size_t)c->nSlotsImg * 16, c->stream));
    return PT_OK;
}

int pt_finish_image(pt_ctx* c, int age) {
    if (!c || age < 0 || age >= pt_ctx::IMAGES) return fail(PT_ERR_ARG, "pt_finish_image: age must be in [0,3] (0 = current image)");
    MULTI_ALL(c, pt_finish_image(k, age));
    HIP_TRY(hipSetDevice(c->device));
    return pump(c, PUMP_IMAGE, (c->curImage + pt_ctx::IMAGES - age) % pt_ctx::IMAGES);
}

int pt_image_device(pt_ctx* c, int age, void** dev_ptr, size_t* n_pixels) {
    if (!c || !dev_ptr || !n_pixels || age < 0 || age >= pt_ctx::IMAGES) return fail(PT_ERR_ARG, "pt_image_device: bad argument");
    if (c->multi) return fail(PT_ERR_ARG, "pt_image_device: a multi-GPU context has one packed accumulator per device; pt_gather_image delivers the whole image");
    float4* img = c->dImage[(c->curImage + pt_ctx::IMAGES - age) % pt_ctx::IMAGES];
    if (!img) return fail(PT_ERR_ARG, "pt_image_device: no image of that age yet (too few pt_next_image calls)");
    *dev_ptr = img; *n_pixels = (size_t)c->nSlotsImg;
    return PT_OK;
}

int pt_synchronize(pt_ctx* c) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    MULTI_ALL(c, pt_synchronize(k));
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = flushStream(c))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_stream_wait(pt_ctx* c) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    if (c->multi) { for (pt_ctx* k : c->multi->kids) { int rc = pt_stream_wait(k); if (rc) return rc; } return PT_OK; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_read_frame(pt_ctx* c, float* out) {
    if (!c || !out) return fail(PT_ERR_ARG, "pt_read_frame: null argument");
    if (c->multi) {                                               // the ONE collective: gather on devices[0], un-tile, read back
        float4* full = nullptr;
        int rc = multiGather(c, 0, &full);
        if (rc) return rc;
        MultiCtx& M = *c->multi;
        pt_ctx* root = M.kids[0];
        if (M.shardTotal == M.n) {
            HIP_TRY(hipMemcpyAsync(out, full, (size_t)c->W * c->H * 16, hipMemcpyDeviceToHost, root->stream));
            HIP_TRY(hipStreamSynchronize(root->stream));
            return PT_OK;
        }
        // a part of the image (pt_create_multi_part): only this group's pixels are written, like a single shard context
        std::vector<float> tmp(M.maps.size() * 4);
        HIP_TRY(hipMemcpyAsync(tmp.data(), full, tmp.size() * 4, hipMemcpyDeviceToHost, root->stream));
        HIP_TRY(hipStreamSynchronize(root->stream));
        for (size_t k = 0; k < M.maps.size(); k++) if (M.maps[k] >= 0) std::memcpy(out + 4 * (size_t)M.maps[k], tmp.data() + 4 * k, 16);
        return PT_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    { int rc; if ((rc = flushStream(c))) return rc; }
    float4* const dFrame = c->dImage[c->curImage];
    if (c->shardCount == 1) {
        HIP_TRY(hipMemcpyAsync(out, dFrame, (size_t)c->W * c->H * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return PT_OK;
    }
    std::vector<float> tmp((size_t)c->nLocal * 4);
    HIP_TRY(hipMemcpyAsync(tmp.data(), dFrame, tmp.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->nLocal; k++) std::memcpy(out + 4 * (size_t)c->pixList[k], tmp.data() + 4 * (size_t)k, 16);
    return PT_OK;
}

/* The inverse of pt_read_frame: FRAME is the path tracer's only persistent state (frag.glsl:924-933: rgb = running sum, a = count), so a saved image written
 * back lets an interrupted accumulation go on — N frames, read, (new context,) write, M frames more = N + M frames, bit for bit. */
int pt_write_frame(pt_ctx* c, const float* in) {
    if (!c || !in) return fail(PT_ERR_ARG, "pt_write_frame: null argument");
    MULTI_ALL(c, pt_write_frame(k, in));                           // every stream takes the pixels of its own tile shard
    HIP_TRY(hipSetDevice(c->device));
    { int rc; if ((rc = flushStream(c))) return rc; }
    float4* const dFrame = c->dImage[c->curImage];
    if (c->shardCount == 1) {
        HIP_TRY(hipMemcpyAsync(dFrame, in, (size_t)c->W * c->H * 16, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return PT_OK;
    }
    std::vector<float> tmp((size_t)c->nSlotsImg * 4, 0.0f);       // shard-local order, the padding slots zero as pt_reset_frame leaves them
    for (int k = 0; k < c->nLocal; k++) std::memcpy(tmp.data() + 4 * (size_t)k, in + 4 * (size_t)c->pixList[k], 16);
    HIP_TRY(hipMemcpyAsync(dFrame, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_read_display(pt_ctx* c, int frame_count, int java_bytes, uint8_t* rgb_out) {
    if (!c || !rgb_out) return fail(PT_ERR_ARG, "pt_read_display: null argument");
    const float4* frame = nullptr; pt_ctx* on = c;
    if (c->multi) {
        if (c->multi->shardTotal != c->multi->n) return fail(PT_ERR_ARG, "pt_read_display needs the whole image: this group holds a part of it (pt_create_multi_part)");
        float4* full = nullptr;
        int rc = multiGather(c, 0, &full);
        if (rc) return rc;
        frame = full; on = c->multi->kids[0];
    } else {
        if (c->shardCount != 1) return fail(PT_ERR_ARG, "pt_read_display needs the whole image: a single shard of several cannot show it (use a pt_create_multi context)");
        HIP_TRY(hipSetDevice(c->device));
        { int rc; if ((rc = flushStream(c))) return rc; }
        frame = c->dImage[c->curImage];
    }
    HIP_TRY(hipSetDevice(on->device));
    const size_t bytes = (size_t)c->W * c->H * 3;
    if (!on->dDisplay) HIP_TRY(hipMalloc((void**)&on->dDisplay, bytes));
    hipLaunchKernelGGL(k_display, dim3((unsigned)(((size_t)c->W * c->H + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, on->stream, frame, c->W, c->H, (float)frame_count, java_bytes, on->dDisplay);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb_out, on->dDisplay, bytes, hipMemcpyDeviceToHost, on->stream));
    HIP_TRY(hipStreamSynchronize(on->stream));
    return PT_OK;
}

namespace {
// minimal PNG writer: IHDR + one IDAT of stored (uncompressed) deflate blocks + IEND
struct Crc32Table {
    uint32_t T[256];
    constexpr Crc32Table() : T{} { for (uint32_t i = 0; i < 256; i++) { uint32_t x = i; for (int k = 0; k < 8; k++) x = (x & 1) ? 0xedb88320u ^ (x >> 1) : x >> 1; T[i] = x; } }
};
constexpr Crc32Table kCrc32{};      // built at compile time: contexts may write PNGs from different threads at once
uint32_t crc32_(const uint8_t* p, size_t n, uint32_t c) {
    c = ~c;
    for (size_t i = 0; i < n; i++) c = kCrc32.T[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}
void pngChunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    auto be32 = [&](uint32_t v) { for (int k = 3; k >= 0; k--) out.push_back((uint8_t)(v >> (8 * k))); };
    be32((uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
    be32(crc32_(out.data() + at, out.size() - at, 0));
}
static std::vector<uint8_t> encodePng(const uint8_t* rgb, int W, int H) {
    std::vector<uint8_t> raw; raw.reserve((size_t)H * (3 * (size_t)W + 1));
    for (int y = 0; y < H; y++) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)y * W * 3, rgb + (size_t)(y + 1) * W * 3); }      // filter type 0 per scanline
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw.size(); i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    for (int k = 3; k >= 0; k--) z.push_back((uint8_t)(((b << 16) | a) >> (8 * k)));
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    for (uint32_t v : {(uint32_t)W, (uint32_t)H}) for (int k = 3; k >= 0; k--) ihdr.push_back((uint8_t)(v >> (8 * k)));
    for (uint8_t v : {8, 2, 0, 0, 0}) ihdr.push_back(v);          // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    pngChunk(out, "IHDR", ihdr); pngChunk(out, "IDAT", z); pngChunk(out, "IEND", {});
    return out;
}
}  // namespace

int pt_save_png(pt_ctx* c, int frame_count, int java_bytes, const char* path) {
    if (!c || !path) return fail(PT_ERR_ARG, "pt_save_png: null argument");
    std::vector<uint8_t> rgb((size_t)c->W * c->H * 3);
    int rc = pt_read_display(c, frame_count, java_bytes, rgb.data());
    if (rc) return rc;
    const std::vector<uint8_t> png = encodePng(rgb.data(), c->W, c->H);
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(PT_ERR_ARG, std::string("pt_save_png: cannot open ") + path);
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    if (std::fclose(f) != 0 || !ok) return fail(PT_ERR_ARG, std::string("pt_save_png: short write to ") + path);
    return PT_OK;
}

/* One image of a multi-GPU context: finish image `age` on every device, ONE RCCL gather of the packed accumulators on device[0],
 * un-tile there (stream-ordered on device[0]'s stream, not synchronised).  A one-device context returns its own image. */
int pt_gather_image(pt_ctx* c, int age, void** full_dev) {
    if (!c || !full_dev || age < 0 || age >= pt_ctx::IMAGES) return fail(PT_ERR_ARG, "pt_gather_image: bad argument");
    if (c->multi) { float4* full = nullptr; int rc = multiGather(c, age, &full); if (rc) return rc; *full_dev = full; return PT_OK; }
    if (c->shardCount != 1) return fail(PT_ERR_ARG, "pt_gather_image: this context is one shard of several; the gather belongs to the multi-GPU context (pt_create_multi) or to the host layer");
    int rc = pt_finish_image(c, age);
    if (rc) return rc;
    *full_dev = c->dImage[(c->curImage + pt_ctx::IMAGES - age) % pt_ctx::IMAGES];
    if (!*full_dev) return fail(PT_ERR_ARG, "pt_gather_image: no image of that age yet (too few pt_next_image calls)");
    return PT_OK;
}

int pt_frame_device(pt_ctx* c, void** dev_ptr, size_t* n_pixels) {
    if (!c || !dev_ptr || !n_pixels) return fail(PT_ERR_ARG, "pt_frame_device: null argument");
    if (c->multi) return fail(PT_ERR_ARG, "pt_frame_device: a multi-GPU context has one packed accumulator per device; pt_gather_image delivers the whole image");
    *dev_ptr = c->dImage[c->curImage]; *n_pixels = (size_t)c->nSlotsImg;
    return PT_OK;
}

int pt_shard_slots(int width, int height, int shard_count, size_t* n_slots) {
    if (width < 1 || height < 1 || shard_count < 1 || !n_slots) return fail(PT_ERR_ARG, "pt_shard_slots: bad argument");
    *n_slots = shard_count == 1 ? (size_t)width * height : shardSlots(width, height, shard_count);
    return PT_OK;
}

int pt_shard_map(int width, int height, int shard_rank, int shard_count, int32_t* out, size_t n_slots) {
    if (width < 1 || height < 1 || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count || !out) return fail(PT_ERR_ARG, "pt_shard_map: bad argument");
    std::vector<int32_t> px;
    if (shard_count == 1) { for (size_t i = 0; i < n_slots; i++) out[i] = i < (size_t)width * height ? (int32_t)i : -1; return PT_OK; }
    shardPixels(width, height, shard_rank, shard_count, px);
    if (px.size() > n_slots) return fail(PT_ERR_ARG, "pt_shard_map: n_slots too small");
    for (size_t i = 0; i < n_slots; i++) out[i] = i < px.size() ? px[i] : -1;
    return PT_OK;
}

int pt_unshard(pt_ctx* c, const void* gathered_dev, void* full_dev) {
    if (!c || !gathered_dev || !full_dev) return fail(PT_ERR_ARG, "pt_unshard: null argument");
    if (c->multi) {
        if (c->multi->shardTotal == c->multi->n) return fail(PT_ERR_ARG, "pt_unshard: a whole-image context gathers and un-tiles by itself (pt_gather_image, pt_read_frame)");
        c = c->multi->kids[0];                                    // a part of the image: any of its shards knows the layout of all shard_total blocks
    }
    HIP_TRY(hipSetDevice(c->device));
    size_t total = (size_t)c->nSlotsImg * c->shardCount;
    if (!c->dAllMaps) {
        std::vector<int32_t> maps(total);
        for (int r = 0; r < c->shardCount; r++) { int rc = pt_shard_map(c->W, c->H, r, c->shardCount, maps.data() + (size_t)r * c->nSlotsImg, (size_t)c->nSlotsImg); if (rc) return rc; }
        HIP_TRY(hipMalloc((void**)&c->dAllMaps, total * 4));
        HIP_TRY(hipMemcpy(c->dAllMaps, maps.data(), total * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_unshard, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, (const float4*)gathered_dev, c->dAllMaps, c->nSlotsImg, c->shardCount, (float4*)full_dev);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_set_stream(pt_ctx* c, void* hip_stream) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    if (c->multi) return fail(PT_ERR_ARG, "pt_set_stream: a multi-GPU context owns one stream per device");
    HIP_TRY(hipSetDevice(c->device));
    { int rc; if ((rc = flushStream(c))) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->ownStream;
    return PT_OK;
}

int pt_set_option(pt_ctx* c, int option, int64_t value) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    MULTI_ALL(c, pt_set_option(k, option, value));
    { int rc; if ((rc = flushStream(c))) return rc; }
    switch (option) {
        case 0: if (value != 0 && (value < BLOCK || value > (1 << 26))) return fail(PT_ERR_ARG, "path slots must be 0 (automatic) or in [256, 2^26]"); c->poolSlots = (int)((value + BLOCK - 1) / BLOCK * BLOCK); return PT_OK;
        case 1: c->countStats = value != 0; return PT_OK;
        case 2: if (value < 0 || value > 160 * 1024) return fail(PT_ERR_ARG, "LDS budget out of range"); c->ldsBudget = (int)value; c->sceneDirty = true; return PT_OK;
        case 3: if (value < 1 || value > 64) return fail(PT_ERR_ARG, "next-object threshold must be in [1,64]"); c->noneMin = (int)value; c->noneMinSet = true; return PT_OK;
        case 4: if (value < 0 || value > 2) return fail(PT_ERR_ARG, "extend mode must be 0, 1 or 2"); c->extendMode = (int)value; return PT_OK;
        case 16: if (value != 0 && value != 1) return fail(PT_ERR_ARG, "numeric contract: 0 exact (bit-identical to the oracle), 1 relaxed (hardware rcp/rsq/sqrt/log/cos; RMSE <= 1e-3)"); c->fastContract = value != 0; return PT_OK;
        case 19: if (value < -1 || value > 1) return fail(PT_ERR_ARG, "node records of the hand-written kernel: -1 automatic, 0 80-B sign-ordered, 1 64-B"); c->asmNodeLayout = (int)value; c->sceneDirty = true; return PT_OK;
        case 18: if (value < 0 || value > 2) return fail(PT_ERR_ARG, "index-stack encoding: 0 automatic, 1 at least 8-bit codes, 2 the floats themselves"); c->forceNiBits8 = (int)value; c->sceneDirty = true; return PT_OK;
        case 21: if (value < 0 || value > 7) return fail(PT_ERR_ARG, "spatial partition: eighths of every XCD's CUs for the intersect kernel (0 = off: both kernels on all CUs)"); c->cuPartition = (int)value; return PT_OK;
        case 20: if (value != 0 && value != 1) return fail(PT_ERR_ARG, "per-ray cull of the object loop (more than 8 BVHs): 0 off, 1 on"); c->asmNoRootCull = value == 0; c->sceneDirty = true; return PT_OK;
        case 17: if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(PT_ERR_ARG, "block size of the hand-written kernel: 0 automatic, 256, 512 or 1024"); c->asmTpb = (int)value; return PT_OK;
        case 14: if (value < -1 || value > 1) return fail(PT_ERR_ARG, "main loop of the hand-written kernel: -1 automatic, 0 phase-voting, 1 fused trip"); c->asmLoop = (int)value; return PT_OK;
        case 13: return c->asmLaunches > (uint64_t)value ? PT_OK : fail(PT_ERR_UNSUPPORTED, "the hand-written intersect kernel has been launched " + std::to_string(c->asmLaunches) + " times");      // query (debug)
        case 12: {                                                // query (debug): 0 = the current scene runs on the hand-written intersect kernel, else PT_ERR_UNSUPPORTED + why not
            if (c->sceneDirty) { int rc = buildScene(c); if (rc) return rc; }
            return c->asmEligible ? PT_OK : fail(PT_ERR_UNSUPPORTED, "compiled intersect kernel: " + c->asmWhyNot);
        }
        case 5: if (value != 64 && value != 128 && value != 256 && value != 512 && value != 1024) return fail(PT_ERR_ARG, "extend block size must be 64, 128, 256, 512 or 1024"); c->extendTpb = (int)value; return PT_OK;
        case 6: if (value < 0 || value > 150 * 1024) return fail(PT_ERR_ARG, "extend LDS cache bytes out of range"); c->extendCacheBytes = (int)value; c->extendCacheSet = true; c->sceneDirty = true; return PT_OK;
        case 7: if (value < 1 || value > 64) return fail(PT_ERR_ARG, "refill threshold must be in [1,64]"); c->refillMin = (int)value; return PT_OK;
        case 8: if (value < 0 || value > 32) return fail(PT_ERR_ARG, "blocks per CU must be in [0,32]"); c->extendMaxBlocksPerCU = (int)value; return PT_OK;
        case 9: if (value < 0 || value > 8) return fail(PT_ERR_ARG, "inner-phase persistence must be in [0,8] eighths"); c->innerKeepEighths = (int)value; return PT_OK;
        case 11: if (value < -1 || value > 2) return fail(PT_ERR_ARG, "stack mode must be -1 (automatic), 0, 1 or 2"); c->stackModeForce = (int)value; c->sceneDirty = true; return PT_OK;
        case 10: if (value < 0 || value > 0x7fffffff) return fail(PT_ERR_ARG, "breadth-first node count out of range"); c->bfsNodes = (int)value; c->sceneDirty = true; return PT_OK;
    }
    return fail(PT_ERR_ARG, "unknown option");
}

int pt_get_counters(pt_ctx* c, uint64_t* out, int n) {
    if (!c || !out) return fail(PT_ERR_ARG, "pt_get_counters: null argument");
    if (c->multi) {                                               // whole-image totals: the sum over the shards
        for (int k = 0; k < n && k < PT_CNT_N; k++) out[k] = 0;
        for (pt_ctx* kid : c->multi->kids) {
            uint64_t one[PT_CNT_N] = {0};
            int rc = pt_get_counters(kid, one, PT_CNT_N);
            if (rc) return rc;
            for (int k = 0; k < n && k < PT_CNT_N; k++) out[k] += one[k];
        }
        return PT_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    { int rc; if ((rc = flushStream(c))) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    Control h;
    HIP_TRY(hipMemcpy(&h, c->dCtl, sizeof(h), hipMemcpyDeviceToHost));
    uint64_t all[PT_CNT_N];
    for (int k = 0; k < PT_CNT_N; k++) all[k] = c->hostCnt[k];
    for (int k = 0; k <= PT_CNT_BOXTESTS; k++) all[k] = h.cnt[k];
    for (int k = 0; k < n && k < PT_CNT_N; k++) out[k] = all[k];
    return PT_OK;
}

int pt_reset_counters(pt_ctx* c) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    MULTI_ALL(c, pt_reset_counters(k));
    HIP_TRY(hipSetDevice(c->device));
    { int rc; if ((rc = flushStream(c))) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemset(c->dCtl->cnt, 0, sizeof(Control::cnt)));
    HIP_TRY(hipMemset(c->dCtl->dbg, 0, sizeof(Control::dbg)));
    std::memset(c->hostCnt, 0, sizeof(c->hostCnt));
    for (auto& k : c->kt) { k.used = 0; k.ms = 0; k.launches = 0; k.each.clear(); }
    return PT_OK;
}

int pt_debug_phase_stats(pt_ctx* c, uint64_t* out, int n) {
    if (!c || !out) return fail(PT_ERR_ARG, "pt_debug_phase_stats: null argument");
    if (c->multi) c = c->multi->kids[0];
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));                     // what the launches so far left; submitted batches are NOT completed first
    Control h;
    HIP_TRY(hipMemcpy(&h, c->dCtl, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < n && k < 16; k++) out[k] = h.dbg[k];
    if (c->dAsmDbg && n >= 16 + 8192 * 8) {                       // developer builds of the hand-written kernel (-DPT_ASM_DEBUG / -DPT_ASM_PROF): 16 words per block of the last launch
        HIP_TRY(hipMemcpy(out + 16, c->dAsmDbg, 8192 * 64, hipMemcpyDeviceToHost));
        return PT_OK;
    }
#if defined(PT_PHASE_STATS) || defined(PT_WAVE_STAMPS)
    for (int k = 16; k < n && k < 16 + 8192; k++) out[k] = h.waveEnd[k - 16];
    for (int k = 16 + 8192; k < n && k < 16 + 2 * 8192; k++) out[k] = h.waveStart[k - 16 - 8192];
#endif
    return PT_OK;
}

int pt_set_timing(pt_ctx* c, int enabled) {
    if (!c) return fail(PT_ERR_ARG, "null context");
    if (c->multi) { for (pt_ctx* k : c->multi->kids) k->timing = enabled != 0; return PT_OK; }
    c->timing = enabled != 0;
    return PT_OK;
}

int pt_kernel_time(pt_ctx* c, int kernel, int64_t* launches, double* total_ms) {
    if (!c || kernel < 0 || kernel > 3 || !launches || !total_ms) return fail(PT_ERR_ARG, "pt_kernel_time: bad argument");
    if (c->multi) {                                               // launches and device time summed over the shards: total / launches = mean launch
        *launches = 0; *total_ms = 0;
        for (pt_ctx* kid : c->multi->kids) { int64_t l; double ms; int rc = pt_kernel_time(kid, kernel, &l, &ms); if (rc) return rc; *launches += l; *total_ms += ms; }
        return PT_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = resolveTimes(c);
    if (rc) return rc;
    *launches = c->kt[kernel].launches; *total_ms = c->kt[kernel].ms;
    return PT_OK;
}

int pt_kernel_time_median(pt_ctx* c, int kernel, double* median_ms) {
    if (!c || kernel < 0 || kernel > 3 || !median_ms) return fail(PT_ERR_ARG, "pt_kernel_time_median: bad argument");
    if (c->multi) c = c->multi->kids[0];
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = resolveTimes(c);
    if (rc) return rc;
    std::vector<float> v = c->kt[kernel].each;
    if (v.empty()) { *median_ms = 0; return PT_OK; }
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    *median_ms = v[v.size() / 2];
    return PT_OK;
}

int pt_debug_math(pt_ctx* c, int fn, const float* x, const float* y, float* out, size_t n) {
    if (!c || !x || !out) return fail(PT_ERR_ARG, "pt_debug_math: null argument");
    if (c->multi) c = c->multi->kids[0];
    HIP_TRY(hipSetDevice(c->device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    Scratch scratch{{(void**)&dx, (void**)&dy, (void**)&dout}};       // freed on every return path
    HIP_TRY(hipMalloc((void**)&dx, n * 4)); HIP_TRY(hipMalloc((void**)&dy, n * 4)); HIP_TRY(hipMalloc((void**)&dout, n * 4));
    HIP_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    if (y) HIP_TRY(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice)); else HIP_TRY(hipMemset(dy, 0, n * 4));
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, fn, dx, dy, dout, n);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_debug_intersect(pt_ctx* c, const float* o, const float* d, float* out, size_t n) {
    if (!c || !o || !d || !out || n < 1 || n > (1u << 24)) return fail(PT_ERR_ARG, "pt_debug_intersect: bad argument");
    if (c->multi) c = c->multi->kids[0];
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = flushStream(c))) return rc;                         // before the frame constants of a running stream are overwritten
    std::memset(&c->streamIn, 0xff, sizeof(FrameIn));             // ... which are no stream's any more afterwards
    if (c->sceneDirty && (rc = buildScene(c))) return rc;
    size_t np = (n + BLOCK - 1) / BLOCK * BLOCK;
    std::vector<float> g0(np * 4, 0.0f), g1(np * 4, 0.0f);
    for (size_t i = 0; i < n; i++) {
        g0[4 * i] = o[3 * i]; g0[4 * i + 1] = o[3 * i + 1]; g0[4 * i + 2] = o[3 * i + 2]; g0[4 * i + 3] = d[3 * i];
        g1[4 * i] = d[3 * i + 1]; g1[4 * i + 1] = d[3 * i + 2]; uint32_t fl = FL_ALIVE; std::memcpy(&g1[4 * i + 3], &fl, 4);
    }
    State st{};
    Scratch scratch{{(void**)&st.G0, (void**)&st.G1, (void**)&st.H}};  // freed on every return path
    HIP_TRY(hipMalloc((void**)&st.G0, np * 16)); HIP_TRY(hipMalloc((void**)&st.G1, np * 16)); HIP_TRY(hipMalloc((void**)&st.H, np * 16));
    HIP_TRY(hipMemcpy(st.G0, g0.data(), np * 16, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(st.G1, g1.data(), np * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(st.H, 0, np * 16, c->stream));       // ordered before the kernels below (the context's stream does not wait for the null stream)
    // the ellipsoid rotation matrices are produced by k_frame_setup
    FrameIn fin; std::memset(&fin, 0, sizeof(fin));
    if (c->params.size() >= 12) std::memcpy(fin.params, c->params.data(), 48);
    fin.params[11] = 0.0f;
    HIP_TRY(hipMemcpy(c->dFrameIn, &fin, sizeof(fin), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_frame_setup, dim3(1), dim3(64), 0, c->stream, c->sc, c->dFrameIn, c->dFc, c->dEllip);
    size_t ldsBytes = (size_t)c->sc.ldsNodes * 64 + (size_t)c->sc.ldsTris * 48 + (size_t)c->stackDepth * BLOCK * 4;
    hipLaunchKernelGGL(k_init_control, dim3(1), dim3(1), 0, c->stream, c->dCtl);
    if (c->extendMode == 0) {
        hipLaunchKernelGGL(k_extend<false>, dim3((unsigned)(np / BLOCK)), dim3(BLOCK), ldsBytes, c->stream, c->sc, st, (const unsigned*)nullptr, 0, (int)np, c->dCtl);
    } else {                                                      // the production kernels (persistent blocks; hand-written or compiled), as pump() launches them
        std::memcpy(c->streamIn.params, fin.params, 48); c->streamIn.params[9] = 1.0f;      // no thickness probes in this pool
        PoolRun pr; pr.stream = c->stream; pr.st = st; pr.launched = (unsigned)np; pr.iter = 0;
        c->debugExactExtend = true;
        TIMED_LAUNCH_ON(c->stream, 0, launchExtendPersist(c, pr));       // (pt_set_timing: scripts/coherence_probe.py times the production kernel on ray sets of its own)
        c->debugExactExtend = false;
        std::memset(&c->streamIn, 0xff, sizeof(FrameIn));
    }
    HIP_TRY(hipGetLastError());
    if (!c->asmError.empty()) { const std::string m = c->asmError; c->asmError.clear(); return fail(PT_ERR_HIP, m); }
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<float> h(np * 4);
    HIP_TRY(hipMemcpy(h.data(), st.H, np * 16, hipMemcpyDeviceToHost));
#ifdef PT_ASM_DEBUG
    if (c->dAsmDbg) {
        std::vector<unsigned> dbg(8192 * 16);
        HIP_TRY(hipMemcpy(dbg.data(), c->dAsmDbg, dbg.size() * 4, hipMemcpyDeviceToHost));
        for (int w = 0; w < 24; w++) { fprintf(stderr, "asm wave %d:", w); for (int k = 0; k < 9; k++) fprintf(stderr, " %u", dbg[16 * w + k]); fprintf(stderr, "\n"); }
    }
#endif
    std::memcpy(out, h.data(), n * 16);
    return PT_OK;
}

}  // extern "C"
