// pt_bvh.hip — the reference's BVH builder on the GPU (second half).
//
// Restates, bit for bit in double precision, what the Java does per object (all citations: /root/reference/src/Main/dispatch.java):
//   BVH(int triIndicesStart, int triIndicesEnd)   :1630-1646   root bounds = GrowToInclude over the object's triangles, first split
//   splitTEST                                     :1647-1721   5 planes per axis at Min + Size*(i+1)/6, first candidate that beats the
//                                                              INHERITED best cost wins, stable partition by
//                                                              centroid < pos, children with their own grown boxes, leaf when
//                                                              size <= 1, depth limit 256, or no candidate beats the parent's cost
//   testSplitOnTEST / cost                        :1722-1752   |half surface area| * count per side, empty side = infinity
//   node ids                                      :1755-1762   creation order = DFS pre-order
// The Java recursion becomes:
//   * nodes with more than SMALL triangles are split level by level: binned box/count accumulation in LDS (6 bins per axis between the
//     5 planes; min/max/count are order-independent, so the result equals the sequential GrowToInclude), one thread per node evaluates
//     the 15 candidates in the reference's order, and ONE global prefix sum per level turns the left/right flags into a stable partition;
//   * every subtree of at most SMALL triangles is finished by one thread (explicit stack), thousands of them in parallel;
//   * DFS pre-order ids need no traversal: every inner node has two children, so
//         id(v) = depth(v) + 2 * (#leaves left of v's first triangle) - (#right turns on the path root -> v),
//     and the leaf order of the triangles is simply the final permutation.
// min/max run on order-preserving 64-bit keys (native integer atomics); -0.0 < +0.0 as in Java's Math.min/max.  NaN coordinates are
// rejected (the CPU mirror handles them the way the JVM would).  Compiled with -ffp-contract=off like the rest of the library.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "../../../include/pt_api.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

int pt_set_error_(int code, const std::string& msg);     // pt_hip.hip

namespace {

constexpr int SMALL = 64;            // subtrees of at most this many triangles are built by one thread
constexpr int CHUNK = 2048;          // triangles per block in the binning pass
constexpr int MAX_BVH_BRANCHES = 256;    // :45
constexpr int OPT = 5;               // OPTIMIZATION_LEVEL :47
typedef unsigned long long u64;

struct Node {
    double bmin[3], bmax[3];
    double cost;                     // the best cost this node's own split has to beat (its parent's winning cost; root: +inf)
    double pos;
    int start, end, left, right, depth, rturns, axis, splitLevel;
};

struct Lists {                       // device-side work lists and counters of the level loop
    int nodeCount, nLargeNext, nChunkNext, nSmall, rootFailed, hasNaN, maxDepth, pad;
};

__device__ __forceinline__ u64 dkey(double d) { u64 b = (u64)__double_as_longlong(d); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ __forceinline__ double dunkey(u64 k) { u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k; return __longlong_as_double((long long)b); }

struct Box {
    double mn[3], mx[3]; int cnt;
    __device__ void clear() { cnt = 0; for (int k = 0; k < 3; k++) { mn[k] = 0; mx[k] = 0; } }
    __device__ void grow(const double* tmin, const double* tmax) {                 // GrowToInclude :1612-1627 (Math.min/max: -0.0 < +0.0)
        if (cnt) {
            for (int k = 0; k < 3; k++) {
                if (dkey(tmin[k]) < dkey(mn[k])) mn[k] = tmin[k];
                if (dkey(tmax[k]) > dkey(mx[k])) mx[k] = tmax[k];
            }
        } else for (int k = 0; k < 3; k++) { mn[k] = tmin[k]; mx[k] = tmax[k]; }
        cnt++;
    }
    __device__ void merge(const Box& o) { if (!o.cnt) return; int c = cnt; grow(o.mn, o.mx); cnt = c + o.cnt; }
    __device__ double cost() const {                                               // cost :1748-1752 on Size = Max - Min
        if (!cnt) return __longlong_as_double(0x7ff0000000000000ll);
        double sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
        double h = (sx * sy + sx * sz) + sy * sz;
        return fabs(h) * (double)cnt;
    }
};

__device__ __forceinline__ double planePos(const Node& v, int axis, int i) {       // :1654-1655
    double splitPercent = ((double)i + 1.0) / ((double)OPT + 1.0);
    return v.bmin[axis] + (v.bmax[axis] - v.bmin[axis]) * splitPercent;
}

// ---- start: identity permutation, NaN check, root bounds (block reduction + key atomics)
__global__ void k_init(const double* tri, int n, int* idx, int* nodeOf, u64* rootKeys, Lists* L) {
    __shared__ u64 sk[6];
    if (threadIdx.x < 3) sk[threadIdx.x] = ~0ull; else if (threadIdx.x < 6) sk[threadIdx.x] = 0ull;
    __syncthreads();
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        idx[p] = p; nodeOf[p] = 0;
        const double* t = tri + 9 * (size_t)p;
        bool bad = false;
        for (int k = 0; k < 9; k++) bad = bad || (t[k] != t[k]);
        if (bad) L->hasNaN = 1;
        for (int k = 0; k < 3; k++) { atomicMin(&sk[k], dkey(t[k])); atomicMax(&sk[3 + k], dkey(t[3 + k])); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&rootKeys[threadIdx.x], sk[threadIdx.x]); else if (threadIdx.x < 6) atomicMax(&rootKeys[threadIdx.x], sk[threadIdx.x]);
}
__global__ void k_root(Node* nodes, const u64* rootKeys, int n, Lists* L, int* largeCur, int2* chunkCur, int* smallRoots, int* nLarge, int* nChunk) {
    Node v;
    for (int k = 0; k < 3; k++) { v.bmin[k] = dunkey(rootKeys[k]); v.bmax[k] = dunkey(rootKeys[3 + k]); }
    v.cost = __longlong_as_double(0x7ff0000000000000ll); v.pos = -1.0;
    v.start = 0; v.end = n; v.left = -1; v.right = -1; v.depth = 0; v.rturns = 0; v.axis = -1; v.splitLevel = -1;
    nodes[0] = v;
    L->nodeCount = 1;
    if (n > SMALL) {
        largeCur[0] = 0; *nLarge = 1;
        int nch = (n + CHUNK - 1) / CHUNK;
        for (int k = 0; k < nch; k++) chunkCur[k] = make_int2(0, k * CHUNK);
        *nChunk = nch;
    } else { smallRoots[0] = 0; L->nSmall = 1; *nLarge = 0; *nChunk = 0; }
}

// ---- level step 1: bins[slot][18][7] = {min keys xyz, max keys xyz, count} for bin (axis, b), b = #planes with pos <= centroid
__global__ void k_clear_bins(u64* bins, int nLarge) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nLarge * 126) return;
    int f = i % 7;
    bins[i] = f < 3 ? ~0ull : 0ull;
}
__global__ void __launch_bounds__(256) k_bin_large(const double* tri, const int* idx, const Node* nodes, const int* largeCur, const int2* chunks, u64* bins) {
    __shared__ u64 sb[126];
    __shared__ double spos[15];
    const int2 ch = chunks[blockIdx.x];
    const Node& v = nodes[largeCur[ch.x]];
    if (threadIdx.x < 126) sb[threadIdx.x] = (threadIdx.x % 7) < 3 ? ~0ull : 0ull;
    if (threadIdx.x < 15) spos[threadIdx.x] = planePos(v, threadIdx.x / 5, threadIdx.x % 5);
    __syncthreads();
    const int end = min(ch.y + CHUNK, v.end);
    for (int p = ch.y + (int)threadIdx.x; p < end; p += 256) {
        const double* t = tri + 9 * (size_t)idx[p];
        u64 k[6];
        for (int j = 0; j < 6; j++) k[j] = dkey(t[j]);
        for (int a = 0; a < 3; a++) {
            const double c = t[6 + a];
            int b = 0;
            for (int i = 0; i < OPT; i++) b += (c < spos[5 * a + i]) ? 0 : 1;       // left of plane i  <=>  i >= b
            u64* s = sb + (6 * a + b) * 7;
            atomicMin(&s[0], k[0]); atomicMin(&s[1], k[1]); atomicMin(&s[2], k[2]);
            atomicMax(&s[3], k[3]); atomicMax(&s[4], k[4]); atomicMax(&s[5], k[5]);
            atomicAdd(&s[6], 1ull);
        }
    }
    __syncthreads();
    if (threadIdx.x < 126) {
        u64 x = sb[threadIdx.x];
        u64* g = bins + (size_t)ch.x * 126 + threadIdx.x;
        int f = threadIdx.x % 7;
        if (f < 3) atomicMin(g, x); else if (f < 6) atomicMax(g, x); else atomicAdd(g, x);
    }
}

// the children of a node that has just been split (both sides are non-empty: an empty side costs infinity)
__device__ void makeChildren(Node* nodes, int vi, int axis, double pos, double cost, const Box& Lb, const Box& Rb, int level, Lists* L,
                             int* largeNext, int2* chunkNext, int* smallRoots, int* leafStart, int* stackOut, int* nStack) {
    Node& v = nodes[vi];
    const int base = atomicAdd(&L->nodeCount, 2);
    v.axis = axis; v.pos = pos; v.splitLevel = level; v.left = base; v.right = base + 1;
    for (int side = 0; side < 2; side++) {
        const Box& b = side ? Rb : Lb;
        Node c;
        for (int k = 0; k < 3; k++) { c.bmin[k] = b.mn[k]; c.bmax[k] = b.mx[k]; }
        c.cost = cost; c.pos = -1.0;
        c.start = side ? v.start + Lb.cnt : v.start; c.end = side ? v.end : v.start + Lb.cnt;
        c.left = -1; c.right = -1; c.depth = v.depth + 1; c.rturns = v.rturns + side; c.axis = -1; c.splitLevel = -1;
        nodes[base + side] = c;
        atomicMax(&L->maxDepth, c.depth);
        const int m = c.end - c.start;
        if (v.depth >= MAX_BVH_BRANCHES || m <= 1) { leafStart[c.start] = 1; continue; }      // :1690-1693 (MAX_TRIS_IN_BVH_LEAF = 1)
        if (stackOut) { stackOut[(*nStack)++] = base + side; continue; }                      // inside a one-thread subtree
        if (m <= SMALL) { smallRoots[atomicAdd(&L->nSmall, 1)] = base + side; continue; }
        const int slot = atomicAdd(&L->nLargeNext, 1);
        largeNext[slot] = base + side;
        const int nch = (m + CHUNK - 1) / CHUNK;
        const int cb = atomicAdd(&L->nChunkNext, nch);
        for (int k = 0; k < nch; k++) chunkNext[cb + k] = make_int2(slot, c.start + k * CHUNK);
    }
}

// ---- level step 2: one thread per large node evaluates the 15 candidates from the bins, in the reference's order
__global__ void k_eval_large(Node* nodes, const int* largeCur, int nLarge, const u64* bins, int level, Lists* L, int* largeNext, int2* chunkNext,
                             int* smallRoots, int* leafStart) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nLarge) return;
    const int vi = largeCur[slot];
    const Node v = nodes[vi];
    double bestCost = v.cost, bestPos = -1.0; int bestAxis = 0;
    Box bestL, bestR; bestL.clear(); bestR.clear();
    for (int a = 0; a < 3; a++) {
        Box bin[6];
        for (int b = 0; b < 6; b++) {
            const u64* g = bins + (size_t)slot * 126 + (6 * a + b) * 7;
            bin[b].cnt = (int)g[6];
            for (int k = 0; k < 3; k++) { bin[b].mn[k] = bin[b].cnt ? dunkey(g[k]) : 0.0; bin[b].mx[k] = bin[b].cnt ? dunkey(g[3 + k]) : 0.0; }
        }
        Box Lb; Lb.clear();
        for (int i = 0; i < OPT; i++) {
            Lb.merge(bin[i]);
            Box Rb; Rb.clear();
            for (int b = i + 1; b < 6; b++) Rb.merge(bin[b]);
            const double c = Lb.cost() + Rb.cost();
            if (c < bestCost) { bestCost = c; bestAxis = a; bestPos = planePos(v, a, i); bestL = Lb; bestR = Rb; }     // :1657-1661
        }
    }
    if (bestPos == -1.0) {                                                          // :1664 the sentinel doubles as "no split" (Q-11)
        if (vi == 0) L->rootFailed = 1; else leafStart[v.start] = 1;
        return;
    }
    makeChildren(nodes, vi, bestAxis, bestPos, bestCost, bestL, bestR, level, L, largeNext, chunkNext, smallRoots, leafStart, nullptr, nullptr);
}

// ---- level step 3: stable partition of every node split at this level: flags -> global exclusive scan -> scatter
__global__ void k_flags(const double* tri, const int* idx, const int* nodeOf, const Node* nodes, int n, int level, int* flags) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const Node& v = nodes[nodeOf[p]];
    int f = 0;
    if (v.splitLevel == level) f = tri[9 * (size_t)idx[p] + 6 + v.axis] < v.pos ? 1 : 0;      // :1669
    flags[p] = f;
}
__global__ void k_scatter(const int* idx, const int* nodeOf, const Node* nodes, int n, int level, const int* flags, const int* scan, int* idx2, int* nodeOf2) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int vi = nodeOf[p];
    const Node& v = nodes[vi];
    if (v.splitLevel != level) { idx2[p] = idx[p]; nodeOf2[p] = vi; return; }
    const int before = scan[p] - scan[v.start];                 // lefts before p inside the node
    const int nLeft = nodes[v.left].end - v.start;
    int q, child;
    if (flags[p]) { q = v.start + before; child = v.left; }
    else { q = v.start + nLeft + ((p - v.start) - before); child = v.right; }
    idx2[q] = idx[p]; nodeOf2[q] = child;
}

// ---- every subtree of <= SMALL triangles: one thread, the reference's loop as written
__global__ void __launch_bounds__(64) k_small(const double* tri, int* idx, int* scratch, Node* nodes, const int* smallRoots, int nSmall, Lists* L, int* leafStart) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nSmall) return;
    int stack[SMALL + 8];
    int sp = 0;
    stack[sp++] = smallRoots[s];
    while (sp > 0) {
        const int vi = stack[--sp];
        const Node v = nodes[vi];
        double bestCost = v.cost, bestPos = -1.0; int bestAxis = 0;
        Box bestL, bestR; bestL.clear(); bestR.clear();
        for (int a = 0; a < 3; a++) {
            for (int i = 0; i < OPT; i++) {
                const double pos = planePos(v, a, i);
                Box Lb, Rb; Lb.clear(); Rb.clear();
                for (int p = v.start; p < v.end; p++) {                              // testSplitOnTEST :1722-1747
                    const double* t = tri + 9 * (size_t)idx[p];
                    if (t[6 + a] < pos) Lb.grow(t, t + 3); else Rb.grow(t, t + 3);
                }
                const double c = Lb.cost() + Rb.cost();
                if (c < bestCost) { bestCost = c; bestAxis = a; bestPos = pos; bestL = Lb; bestR = Rb; }
            }
        }
        if (bestPos == -1.0) {
            if (vi == 0) L->rootFailed = 1; else leafStart[v.start] = 1;
            continue;
        }
        int q = v.start;                                                              // stable partition through the scratch segment
        for (int p = v.start; p < v.end; p++) { int t = idx[p]; if (tri[9 * (size_t)t + 6 + bestAxis] < bestPos) scratch[q++] = t; }
        for (int p = v.start; p < v.end; p++) { int t = idx[p]; if (!(tri[9 * (size_t)t + 6 + bestAxis] < bestPos)) scratch[q++] = t; }
        for (int p = v.start; p < v.end; p++) idx[p] = scratch[p];
        makeChildren(nodes, vi, bestAxis, bestPos, bestCost, bestL, bestR, 1 << 30, L, nullptr, nullptr, nullptr, leafStart, stack, &sp);
    }
}

// ---- numbering and output
__global__ void k_emit(const Node* nodes, int nNodes, const int* leafScan, double* outBounds, int32_t* outLinks, int32_t* outLeaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes) return;
    const Node& v = nodes[i];
    auto idOf = [&](const Node& x) { return x.depth + 2 * leafScan[x.start] - x.rturns; };
    const int id = idOf(v);
    for (int k = 0; k < 3; k++) { outBounds[6 * (size_t)id + k] = v.bmin[k]; outBounds[6 * (size_t)id + 3 + k] = v.bmax[k]; }
    if (v.left >= 0) {
        outLinks[2 * (size_t)id] = idOf(nodes[v.left]); outLinks[2 * (size_t)id + 1] = idOf(nodes[v.right]);
        outLeaf[2 * (size_t)id] = 0; outLeaf[2 * (size_t)id + 1] = 0;
    } else {
        outLinks[2 * (size_t)id] = -1; outLinks[2 * (size_t)id + 1] = -1;
        outLeaf[2 * (size_t)id] = v.start; outLeaf[2 * (size_t)id + 1] = v.end;
    }
}

#define BVH_TRY(x)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) { rc = pt_set_error_(PT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); goto done; } \
    } while (0)

}  // namespace

extern "C" int pt_build_bvh(int device, const double* tri9, int64_t n_tris, int32_t* n_nodes, double* node_bounds, int32_t* node_links,
                            int32_t* node_leaf, int32_t* leaf_tris, int32_t* max_depth) {
    if (!tri9 || !n_nodes || !node_bounds || !node_links || !node_leaf || !leaf_tris || n_tris < 1 || n_tris > (1ll << 30))
        return pt_set_error_(PT_ERR_ARG, "pt_build_bvh: bad argument");
    int rc = PT_OK;
    const int n = (int)n_tris;
    const int maxNodes = 2 * n;                 // a full binary tree over at most n leaves
    const int maxLarge = n / SMALL + 2, maxChunks = n / CHUNK + maxLarge + 2;
    double* dTri = nullptr; int *dIdx[2] = {nullptr, nullptr}, *dNodeOf[2] = {nullptr, nullptr}, *dFlags = nullptr, *dScan = nullptr, *dLeafStart = nullptr;
    Node* dNodes = nullptr; u64 *dBins = nullptr, *dRootKeys = nullptr; Lists* dL = nullptr; int *dLarge[2] = {nullptr, nullptr}, *dSmall = nullptr, *dCounts = nullptr;
    int2* dChunk[2] = {nullptr, nullptr};
    void* dTemp = nullptr; size_t tempBytes = 0;
    double* dOutB = nullptr; int32_t *dOutLinks = nullptr, *dOutLeaf = nullptr;
    hipStream_t s = nullptr;
    Lists h; int counts[2]; int cur = 0, level = 0, nLarge = 0, nChunk = 0;
    const int gridN = (n + 255) / 256;
    u64 rk[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};

    BVH_TRY(hipSetDevice(device));
    BVH_TRY(hipStreamCreate(&s));
    BVH_TRY(hipMalloc((void**)&dTri, (size_t)n * 72));
    for (int k = 0; k < 2; k++) {
        BVH_TRY(hipMalloc((void**)&dIdx[k], (size_t)n * 4)); BVH_TRY(hipMalloc((void**)&dNodeOf[k], (size_t)n * 4));
        BVH_TRY(hipMalloc((void**)&dLarge[k], (size_t)maxLarge * 4)); BVH_TRY(hipMalloc((void**)&dChunk[k], (size_t)maxChunks * 8));
    }
    BVH_TRY(hipMalloc((void**)&dFlags, (size_t)(n + 1) * 4)); BVH_TRY(hipMalloc((void**)&dScan, (size_t)(n + 1) * 4));
    BVH_TRY(hipMalloc((void**)&dLeafStart, (size_t)(n + 1) * 4));
    BVH_TRY(hipMalloc((void**)&dNodes, (size_t)maxNodes * sizeof(Node)));
    BVH_TRY(hipMalloc((void**)&dBins, (size_t)maxLarge * 126 * 8));
    BVH_TRY(hipMalloc((void**)&dRootKeys, 48)); BVH_TRY(hipMalloc((void**)&dL, sizeof(Lists))); BVH_TRY(hipMalloc((void**)&dSmall, (size_t)(n + 2) * 4));
    BVH_TRY(hipMalloc((void**)&dCounts, 8));
    BVH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, dFlags, dScan, n + 1, s));
    BVH_TRY(hipMalloc(&dTemp, tempBytes));
    BVH_TRY(hipMemcpyAsync(dTri, tri9, (size_t)n * 72, hipMemcpyHostToDevice, s));
    BVH_TRY(hipMemcpyAsync(dRootKeys, rk, 48, hipMemcpyHostToDevice, s));
    BVH_TRY(hipMemsetAsync(dL, 0, sizeof(Lists), s));
    BVH_TRY(hipMemsetAsync(dLeafStart, 0, (size_t)(n + 1) * 4, s));
    hipLaunchKernelGGL(k_init, dim3(gridN), dim3(256), 0, s, dTri, n, dIdx[0], dNodeOf[0], dRootKeys, dL);
    hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, s, dNodes, dRootKeys, n, dL, dLarge[0], dChunk[0], dSmall, dCounts, dCounts + 1);
    BVH_TRY(hipMemcpyAsync(counts, dCounts, 8, hipMemcpyDeviceToHost, s));
    BVH_TRY(hipMemcpyAsync(&h, dL, sizeof(Lists), hipMemcpyDeviceToHost, s));
    BVH_TRY(hipStreamSynchronize(s));
    if (h.hasNaN) { rc = pt_set_error_(PT_ERR_SCENE, "pt_build_bvh: NaN coordinate (build this object with the CPU builder)"); goto done; }
    nLarge = counts[0]; nChunk = counts[1];
    // ---- level loop over the nodes that are still larger than SMALL (one host look per level)
    while (nLarge > 0) {
        if (level > MAX_BVH_BRANCHES + 2) { rc = pt_set_error_(PT_ERR_HIP, "pt_build_bvh: level loop did not terminate (internal error)"); goto done; }
        hipLaunchKernelGGL(k_clear_bins, dim3((nLarge * 126 + 255) / 256), dim3(256), 0, s, dBins, nLarge);
        hipLaunchKernelGGL(k_bin_large, dim3(nChunk), dim3(256), 0, s, dTri, dIdx[cur], dNodes, dLarge[cur], dChunk[cur], dBins);
        hipLaunchKernelGGL(k_eval_large, dim3((nLarge + 63) / 64), dim3(64), 0, s, dNodes, dLarge[cur], nLarge, dBins, level, dL, dLarge[cur ^ 1], dChunk[cur ^ 1],
                           dSmall, dLeafStart);
        hipLaunchKernelGGL(k_flags, dim3(gridN), dim3(256), 0, s, dTri, dIdx[cur], dNodeOf[cur], dNodes, n, level, dFlags);
        BVH_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp, tempBytes, dFlags, dScan, n + 1, s));
        hipLaunchKernelGGL(k_scatter, dim3(gridN), dim3(256), 0, s, dIdx[cur], dNodeOf[cur], dNodes, n, level, dFlags, dScan, dIdx[cur ^ 1], dNodeOf[cur ^ 1]);
        BVH_TRY(hipMemcpyAsync(&h, dL, sizeof(Lists), hipMemcpyDeviceToHost, s));
        BVH_TRY(hipStreamSynchronize(s));
        if (h.rootFailed) break;
        nLarge = h.nLargeNext; nChunk = h.nChunkNext;
        BVH_TRY(hipMemsetAsync(&dL->nLargeNext, 0, 8, s));                      // nLargeNext, nChunkNext
        cur ^= 1; level++;
    }
    if (!h.rootFailed && h.nSmall > 0) {
        hipLaunchKernelGGL(k_small, dim3((h.nSmall + 63) / 64), dim3(64), 0, s, dTri, dIdx[cur], dIdx[cur ^ 1], dNodes, dSmall, h.nSmall, dL, dLeafStart);
        BVH_TRY(hipMemcpyAsync(&h, dL, sizeof(Lists), hipMemcpyDeviceToHost, s));
        BVH_TRY(hipStreamSynchronize(s));
    }
    if (h.rootFailed) {
        rc = pt_set_error_(PT_ERR_SCENE, "BVH root could not be split by any candidate plane (reference: IndexOutOfBoundsException at dispatch.java:1644, SURVEY Q-16)");
        goto done;
    }
    // ---- DFS pre-order ids from the leaf prefix sum; outputs
    BVH_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp, tempBytes, dLeafStart, dScan, n + 1, s));
    BVH_TRY(hipMalloc((void**)&dOutB, (size_t)h.nodeCount * 48)); BVH_TRY(hipMalloc((void**)&dOutLinks, (size_t)h.nodeCount * 8)); BVH_TRY(hipMalloc((void**)&dOutLeaf, (size_t)h.nodeCount * 8));
    hipLaunchKernelGGL(k_emit, dim3((h.nodeCount + 255) / 256), dim3(256), 0, s, dNodes, h.nodeCount, dScan, dOutB, dOutLinks, dOutLeaf);
    BVH_TRY(hipMemcpyAsync(node_bounds, dOutB, (size_t)h.nodeCount * 48, hipMemcpyDeviceToHost, s));
    BVH_TRY(hipMemcpyAsync(node_links, dOutLinks, (size_t)h.nodeCount * 8, hipMemcpyDeviceToHost, s));
    BVH_TRY(hipMemcpyAsync(node_leaf, dOutLeaf, (size_t)h.nodeCount * 8, hipMemcpyDeviceToHost, s));
    BVH_TRY(hipMemcpyAsync(leaf_tris, dIdx[cur], (size_t)n * 4, hipMemcpyDeviceToHost, s));
    BVH_TRY(hipStreamSynchronize(s));
    BVH_TRY(hipGetLastError());
    *n_nodes = h.nodeCount;
    if (max_depth) *max_depth = h.maxDepth;
done:
    {
        void* ptrs[] = {dTri, dIdx[0], dIdx[1], dNodeOf[0], dNodeOf[1], dLarge[0], dLarge[1], dChunk[0], dChunk[1], dFlags, dScan, dLeafStart, dNodes, dBins,
                        dRootKeys, dL, dSmall, dCounts, dTemp, dOutB, dOutLinks, dOutLeaf};
        for (void* p : ptrs) if (p) hipFree(p);
        if (s) hipStreamDestroy(s);
    }
    return rc;
}
