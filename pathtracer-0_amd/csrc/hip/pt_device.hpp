// pt_device.hpp — device-side data layout and the per-lane building blocks of the wavefront
// path tracer (gfx950).  The kernels that schedule them live in pt_hip.hip.
//
// Reference semantics restated here (all citations into /root/reference/src/shaders/frag.glsl):
//   rayBox :408-419, rayTri :351-372, rayEllipsoid :373-384, rayBVH :452-537, rayScene :548-653,
//   RNG :686-708, fresnelReflectAmount :726-743, chooseRay :745-809, trace loop body :820-880,
//   sample set-up in main :894-910, bgCol :235-242, index stack :136-158.
#pragma once
#include "pt_math.hpp"

namespace ptd {
using namespace pm;

// ------------------------------------------------------------------------------------------------
// Device-private scene layout (built on the host from the reference's SSBO contents; traversal
// ORDER is untouched, only the bytes move — SURVEY.md Q-12).
//
//  inner node  = 4 x float4 (64 B): both children's boxes + both child references, so one record
//                per node visit replaces BVHtree[3n+1..2] + 2 x BVHdata rows of the reference:
//                  q0 = (lmin.x, rmin.x, lmin.y, rmin.y)  q1 = (lmin.z, rmin.z, lmax.x, rmax.x)
//                  q2 = (lmax.y, rmax.y, lmax.z, rmax.z)  q3 = (lref, rref, 0, 0) as int bits
//                (left/right interleaved: each register pair feeds one packed-f32 subtract / multiply of rayBox2)
//                child reference: >= 0 inner-node index (BFS order over all objects, so the top
//                levels are the first records = the LDS-staged tile); < 0 leaf whose first
//                triangle record is -(ref+1); REF_EMPTY = leaf without triangles.
//  triangle    = 3 x float4 (48 B) in LEAF order (leafTriIndices already applied):
//                  t0 = v1.xyz, e1.x   t1 = e1.yz, e2.xy   t2 = e2.z, id|last<<31, 0, 0
//                e1 = v2-v1, e2 = v3-v1 are the same binary32 subtractions rayTri performs.
//  shading rec = 4 x float4 (64 B) indexed by triangle id: n1, n2, vt1, vt2, vt3, material
//                (the part of the 160-B reference triangle that only the winning hit needs).
// ------------------------------------------------------------------------------------------------
constexpr int REF_EMPTY = (int)0x80000000;
constexpr int PRIM_NONE = -1;
constexpr int PRIM_ELLIPSOID = 0x40000000;

struct ObjRoot { float bmin[3]; float bmax[3]; int ref; int pad; };           // 32 B
struct EllipRec {                                                                // frag.glsl:606-631
    float c[3], st[3], r; int mat; int rotated; float rot[3]; float R[9]; float RB[9]; float pad[2];
};
struct MatRec {                                                                  // the mtl fields trace()/chooseRay()/directDiffuse() read
    float Kd[3], Ks[3], Ke[3], Tf[3]; float Tr, Ni, Density, Pm, Pr, Pc, Pcr, subsurface; int illum;
    float Ka[3], ssColor[3], ssRadius[3];                                        // only directDiffuse (frag.glsl:661-675)
    int hasMaps;                                                                 // any of the map_* below > -1
    int map_Ka, map_Kd, map_Ks, map_Ke, map_Tr, map_Pm, map_Pr, map_Pc, map_norm; // texture indices (mapMtl :210-225, :827); -1 = none
    int niCode;                                                                  // Ni as an entry of the scene's refraction-index dictionary (DevScene::ni8 / niTable)
};                                                                               // 41 dwords = 164 B
struct TexRec { const uchar4* data; int w, h; };                                 // one entry of the bindless table (binding 15), RGBA8 texels as uploaded

struct FrameConst {            // uniform per batch; written by k_frame_setup
    float screenSize, focalLength, resolution, screenHratio, SAMPLE_RES, MAX_BOUNCES, BLUR, FOCAL_DISTANCE, AUTO_FOCUS;
    float origin[3], rotation[3], mouse[3];
    float camRot[9];           // rotationMatrix(ROTATION), row-major math matrix
    float focus;               // internal_focal_distance of frag.glsl:900-906 (auto-focus ray hoisted)
    float midToScene;
};

struct DevScene {
    const float4* nodes;  int nNodes;        // inner nodes
    const float4* tris;   int nTriRecs;      // leaf-ordered triangle records
    const float4* shade;  int nTris;         // by triangle id
    const int* triObj;                       // by triangle id: index of the object (BVH) whose leaves hold it; -1 none, -2 several
    const ObjRoot* roots; int numObj;
    const EllipRec* ellip; int numEllip;
    const MatRec* mats;   int numMat;
    const uchar4* sky;    int skyW, skyH;   // texture 0 as the RGBA8 texels that were uploaded (dispatch.java:349-354); unorm8() at fetch
    const TexRec* tex;    int numTex;       // the whole texture table (entry 0 = sky again)
    int ldsNodes, ldsTris;                   // how many leading node / triangle records the intersect kernel stages in LDS
    // The refraction-index stack (frag.glsl:136-158) only ever holds 0.0 (an untouched slot), 1.0029 (:816) and the Ni of a material (:834): the path
    // state carries its ten slots as CODES into this dictionary — 3 bits each when the scene has at most 8 distinct values (one dword beside the
    // throughput), 8 bits each otherwise (a 16-B group) — instead of ten floats.  Codes 0 and 1 are 0.0f and 1.0029f.
    float ni8[8];                            // the dictionary of a 3-bit scene (kernel arguments: scalar registers)
    const float* niTable;                    // the dictionary in memory (8-bit scenes)
};

struct Counters { unsigned nodes = 0, tritests = 0, hitupd = 0, boxtests = 0; };

// ------------------------------------------------------------------------------------------------
// Intersection
// ------------------------------------------------------------------------------------------------
PM_DEV float rayBox(vec3 o, vec3 invD, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    float tminx = (mnx - o.x) * invD.x, tminy = (mny - o.y) * invD.y, tminz = (mnz - o.z) * invD.z;
    float tmaxx = (mxx - o.x) * invD.x, tmaxy = (mxy - o.y) * invD.y, tmaxz = (mxz - o.z) * invD.z;
    float t1x = minnum(tminx, tmaxx), t1y = minnum(tminy, tmaxy), t1z = minnum(tminz, tmaxz);
    float t2x = maxnum(tminx, tmaxx), t2y = maxnum(tminy, tmaxy), t2z = maxnum(tminz, tmaxz);
    float tNear = maxnum(maxnum(t1x, t1y), t1z);
    float tFar = minnum(minnum(t2x, t2y), t2z);
    return (tFar >= tNear && tFar > 0.0f) ? (tNear > 0.0f ? tNear : 0.0f) : 1e30f;
}

// rayBox for the two children of a node at once.  Same binary32 operations per box as rayBox; the slab
// subtractions and multiplications of the left and the right box are issued as packed-f32 pairs
// (v_pk_add_f32 / v_pk_mul_f32: two IEEE results per instruction — the kernel is VALU-issue bound).
typedef float f32x2 __attribute__((ext_vector_type(2)));
PM_DEV void rayBox2(vec3 o, vec3 invD, float4 q0, float4 q1, float4 q2, float& Ld, float& Rd) {
    f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    f32x2 ix = {invD.x, invD.x}, iy = {invD.y, invD.y}, iz = {invD.z, invD.z};
    f32x2 tminx = (f32x2{q0.x, q0.y} - ox) * ix, tminy = (f32x2{q0.z, q0.w} - oy) * iy, tminz = (f32x2{q1.x, q1.y} - oz) * iz;
    f32x2 tmaxx = (f32x2{q1.z, q1.w} - ox) * ix, tmaxy = (f32x2{q2.x, q2.y} - oy) * iy, tmaxz = (f32x2{q2.z, q2.w} - oz) * iz;
    {
        float t1x = minnum(tminx.x, tmaxx.x), t1y = minnum(tminy.x, tmaxy.x), t1z = minnum(tminz.x, tmaxz.x);
        float t2x = maxnum(tminx.x, tmaxx.x), t2y = maxnum(tminy.x, tmaxy.x), t2z = maxnum(tminz.x, tmaxz.x);
        float tNear = maxnum(maxnum(t1x, t1y), t1z), tFar = minnum(minnum(t2x, t2y), t2z);
        Ld = (tFar >= tNear && tFar > 0.0f) ? (tNear > 0.0f ? tNear : 0.0f) : 1e30f;
    }
    {
        float t1x = minnum(tminx.y, tmaxx.y), t1y = minnum(tminy.y, tmaxy.y), t1z = minnum(tminz.y, tmaxz.y);
        float t2x = maxnum(tminx.y, tmaxx.y), t2y = maxnum(tminy.y, tmaxy.y), t2z = maxnum(tminz.y, tmaxz.y);
        float tNear = maxnum(maxnum(t1x, t1y), t1z), tFar = minnum(minnum(t2x, t2y), t2z);
        Rd = (tFar >= tNear && tFar > 0.0f) ? (tNear > 0.0f ? tNear : 0.0f) : 1e30f;
    }
}

// Moeller-Trumbore exactly as rayTri; returns 1e30 in t on a miss
PM_DEV void rayTri(vec3 o, vec3 d, vec3 v1, vec3 e1, vec3 e2, float& t, float& u, float& v) {
    const float EPSILON = 1e-10f;
    t = 1e30f; u = 0.0f; v = 0.0f;
    vec3 dCross_e2 = cross(d, e2);
    float det = dot(e1, dCross_e2);
    if (__builtin_fabsf(det) < EPSILON) return;
    float invDet = 1.0f / det;
    vec3 s = o - v1;
    float uu = dot(s, dCross_e2) * invDet;
    if (uu < 0.0f || uu > 1.0f) return;
    vec3 sCross_e1 = cross(s, e1);
    float vv = dot(d, sCross_e1) * invDet;
    if (vv < 0.0f || uu + vv > 1.0f) return;
    float tt = dot(e2, sCross_e1) * invDet;
    if (tt > EPSILON) { t = tt; u = uu; v = vv; }
}

PM_DEV float rayEllipsoid(vec3 o, vec3 d, vec3 c, float r, float f, float g, float h) {
    vec3 oc = o - c;
    float a = f * d.x * d.x + g * d.y * d.y + h * d.z * d.z;
    float b = 2.0f * (f * oc.x * d.x + g * oc.y * d.y + h * oc.z * d.z);
    float C = f * oc.x * oc.x + g * oc.y * oc.y + h * oc.z * oc.z - r * r;
    float Disc = b * b - 4.0f * a * C;
    float sq = __builtin_sqrtf(Disc);
    float t = (sq - b) / (2.0f * a);
    float tAlt = (-b - sq) / (2.0f * a);
    if ((Disc > 0.0f && (tAlt > 0.0f)) || (t > 0.0f)) return (t > tAlt ? tAlt : t);
    return 1e30f;
}

PM_DEV vec3 vecmat(vec3 p, const float* M) {   // p * M, M row-major: component j = dot(p, column j)
    return v3(dot(p, v3(M[0], M[3], M[6])), dot(p, v3(M[1], M[4], M[7])), dot(p, v3(M[2], M[5], M[8])));
}


// Record fetches.  The staged prefix lives in LDS, the rest in global memory; the two loads must stay two
// instructions (ds_read_b128 / global_load_dwordx4).  Left alone, the compiler merges them into ONE flat_load
// through a selected generic pointer, which sends even the LDS hits through the texture addresser — the empty
// asm pins the LDS arm.
PM_DEV void loadNode(const DevScene& sc, const float4* ldsN, int ref, float4& q0, float4& q1, float4& q2, float4& q3) {
    if (ref < sc.ldsNodes) {
        const float4* p = ldsN + 4 * ref;
        q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
        asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x));
    } else {
        const float4* p = sc.nodes + 4 * (size_t)ref;
        q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
    }
}
// The same fetch with both arms issued before either is waited for.  A wave whose lanes sit partly in the staged prefix and partly
// beyond it runs both arms; the compiler gives them the same destination registers and therefore waits for the global loads of the
// other lanes (vmcnt(0)) before it even issues the LDS reads — the two latencies add up.  Written as instructions, the two arms write
// disjoint lanes of the same registers (memory returns are per lane; there is no hazard), and ONE wait covers both.
typedef float f32x4 __attribute__((ext_vector_type(4)));
PM_DEV void loadNode2(const DevScene& sc, const float4* ldsN, int ref, float4& q0, float4& q1, float4& q2, float4& q3) {
    f32x4 a, b, c; f32x2 d;
    asm volatile("" : "=v"(a), "=v"(b), "=v"(c), "=v"(d));            // one set of registers for both arms, defined by whichever arm the lane takes
    if (ref < sc.ldsNodes) {
        const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) float4*)ldsN + 64u * (unsigned)ref;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b64 %3, %4 offset:48"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(addr) : "memory");
    } else {
        const float4* p = sc.nodes + 4 * (size_t)ref;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx2 %3, %4, off offset:48"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
    q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w); q2 = make_float4(c.x, c.y, c.z, c.w); q3 = make_float4(d.x, d.y, 0.0f, 0.0f);
}
PM_DEV void loadTri(const DevScene& sc, const float4* ldsT, int ti, float4& t0, float4& t1, float4& t2) {
    if (ti < sc.ldsTris) {
        const float4* p = ldsT + 3 * ti;
        t0 = p[0]; t1 = p[1]; t2 = p[2];
        asm volatile("" : "+v"(t0.x), "+v"(t1.x), "+v"(t2.x));
    } else {
        const float4* p = sc.tris + 3 * (size_t)ti;
        t0 = p[0]; t1 = p[1]; t2 = p[2];
    }
}

// rayScene's closest-hit search (frag.glsl:548-631) for one ray.  `stk`/`stride`: this lane's
// traversal stack (LDS).  `ldsN`/`ldsT`: LDS copies of the first sc.ldsNodes / sc.ldsTris records.
// Visit order, pruning tests and tie-breaks are the reference's (push far child first; prune
// only at push time; strict '<' on hits), so counters equal the oracle's.
template <bool COUNT>
PM_DEV void intersectScene(const DevScene& sc, vec3 oIn, vec3 d, int* stk, int stride, const float4* ldsN, const float4* ldsT,
                           float& outT, float& outU, float& outV, int& outPrim, Counters& cnt, bool probe = false, int probeObj = 0, float4* hx = nullptr) {
    vec3 o = probe ? oIn : madd(d, 1e-4f, oIn);                    // o = o + 1e-4*d  (:549); the thickness probe calls rayBVH directly (:668)
    vec3 invD = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float closest = 1e30f, hu = 0.0f, hv = 0.0f;
    int prim = PRIM_NONE;
    const int obFirst = probe ? probeObj : 0, obEnd = probe ? probeObj + 1 : sc.numObj;
    for (int ob = obFirst; ob < obEnd; ob++) {
        const ObjRoot R = sc.roots[ob];
        if (COUNT) cnt.boxtests++;
        if (rayBox(o, invD, R.bmin[0], R.bmin[1], R.bmin[2], R.bmax[0], R.bmax[1], R.bmax[2]) > closest) continue;   // :468
        if (R.ref == REF_EMPTY) { if (COUNT) cnt.nodes++; continue; }     // a root without triangles is visited and finds nothing
        int sp = 0;
        stk[0] = R.ref; sp = 1;
        while (sp > 0) {
            int ref = stk[(--sp) * stride];
            if (COUNT) cnt.nodes++;
            if (ref >= 0) {
                float4 q0, q1, q2, q3;
                loadNode(sc, ldsN, ref, q0, q1, q2, q3);
                if (COUNT) cnt.boxtests += 2;
                float Ld, Rd;
                rayBox2(o, invD, q0, q1, q2, Ld, Rd);
                int lref = __float_as_int(q3.x), rref = __float_as_int(q3.y);
                bool pl = Ld < closest, pr = Rd < closest;
                if (COUNT) { if (pl && lref == REF_EMPTY) cnt.nodes++; if (pr && rref == REF_EMPTY) cnt.nodes++; }
                pl = pl && lref != REF_EMPTY; pr = pr && rref != REF_EMPTY;
                if (Ld > Rd) {                                      // :525-531: far child first, near child popped first
                    if (pl) { stk[sp * stride] = lref; sp++; }
                    if (pr) { stk[sp * stride] = rref; sp++; }
                } else {
                    if (pr) { stk[sp * stride] = rref; sp++; }
                    if (pl) { stk[sp * stride] = lref; sp++; }
                }
            } else {
                int ti = -(ref + 1);
                bool last;
                do {
                    float4 t0, t1, t2;
                    loadTri(sc, ldsT, ti, t0, t1, t2);
                    unsigned idl = __float_as_uint(t2.y);
                    last = (idl >> 31) != 0;
                    float t, u, v;
                    if (COUNT) cnt.tritests++;
                    rayTri(o, d, v3(t0.x, t0.y, t0.z), v3(t0.w, t1.x, t1.y), v3(t1.z, t1.w, t2.x), t, u, v);
                    if (t > 0.0f && t < closest) {                  // :489
                        closest = t; hu = u; hv = v; prim = (int)(idl & 0x7fffffffu);
                        if (COUNT) cnt.hitupd++;
                    }
                    ti++;
                } while (!last);
            }
        }
    }
    for (int i = 0; i < (probe ? 0 : sc.numEllip); i++) {           // :606-631
        const EllipRec& E = sc.ellip[i];
        vec3 c = v3(E.c[0], E.c[1], E.c[2]);
        float t;
        if (E.rotated) t = rayEllipsoid(vecmat(o, E.R), vecmat(d, E.R), c, E.r, E.st[0], E.st[1], E.st[2]);
        else t = rayEllipsoid(o, d, c, E.r, E.st[0], E.st[1], E.st[2]);
        if (t < closest) {                                          // hit.parentID keeps the BVH of the last triangle hit (:573): park that
            if (!(prim & PRIM_ELLIPSOID) || prim == PRIM_NONE) {
                // ... and hitUV keeps that triangle's uv (:574 is not undone by :619-630): scenes whose ellipsoids carry texture-mapped
                // materials get the triangle's (u, v, id) in a side record, mapMtl (:826) samples there
                if (hx) *hx = make_float4(hu, hv, __int_as_float(prim), 0.0f);
                hu = __int_as_float(prim);                          // triangle id in u (unused for ellipsoids)
            }
            closest = t; prim = PRIM_ELLIPSOID | i;
        }
    }
    outT = closest; outU = hu; outV = hv; outPrim = prim;
}

// ------------------------------------------------------------------------------------------------
// RNG (PCG hash), frag.glsl:686-708
// ------------------------------------------------------------------------------------------------
PM_DEV uint32_t NextRandom(uint32_t& state) {
    state = state * 747796405u + 2891336453u;
    uint32_t result = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    result = (result >> 22u) ^ result;
    return result;
}
PM_DEV float random_(uint32_t& state) { return (float)NextRandom(state) / 4294967295.0f; }
template <bool FAST = false>
PM_DEV float randValNormalDist(uint32_t& st) {
    const float u1 = random_(st);                  // theta's draw first, then rho's (:697-698)
    const float u2 = random_(st);
    return boxMullerT<FAST>(u1, u2);
}
template <bool FAST = false>
PM_DEV vec3 randLambertianDistVec(uint32_t& st) {
    float x = randValNormalDist<FAST>(st), y = randValNormalDist<FAST>(st), z = randValNormalDist<FAST>(st);
    return v3(x, y, z);
}
// The six draws of randLambertianDistVec when nobody reads the vector (chooseRay draws it before trace() finds the bounce budget used up or the path cut off,
// :775-804 then :866 / :820; the sample loop's next iteration goes on with the advanced rngState, :898-908): six steps of NextRandom's LCG in one
constexpr uint32_t lcgPow(uint32_t a, int n) { return n == 0 ? 1u : a * lcgPow(a, n - 1); }
constexpr uint32_t lcgSum(uint32_t a, uint32_t c, int n) { return n == 0 ? 0u : lcgSum(a, c, n - 1) * a + c; }
PM_DEV uint32_t skipSixDraws(uint32_t state) {
    constexpr uint32_t A6 = lcgPow(747796405u, 6), C6 = lcgSum(747796405u, 2891336453u, 6);
    static_assert(A6 == 1226092713u && C6 == 3109587930u, "six steps of state * 747796405 + 2891336453 (mod 2^32)");
    return state * A6 + C6;
}

// ------------------------------------------------------------------------------------------------
// Path state of one lane (registers); stored SoA in groups of float4 (see pt_hip.hip)
// ------------------------------------------------------------------------------------------------
// flags word of a path slot (G1.w): bits 0-11 bounce, 12-22 sample (the loop counters of frag.glsl:820 / :898; their bounds are
// floats in the shader, here up to 4095 / 2047), 23 FL_INCNZ, 24-27 size of the refraction-index stack (0..10), then the booleans
constexpr uint32_t FL_COUNT_MASK = 0xfffu, FL_SAMPLE_MASK = 0x7ffu; constexpr int FL_SAMPLE_SHIFT = 12, FL_STACK_SHIFT = 24;
// bit 23: incLight of the running sample is not +0.0 in every component, i.e. the group G3 holds it (k_shade reads and writes G3 for such lanes only:
// incLight leaves zero only at an emitter hit that does not end the sample, frag.glsl:865); the sample counter keeps bits 12-22 (SAMPLE_RES <= 2047)
constexpr uint32_t FL_INCNZ = 1u << 23;
constexpr uint32_t FL_INOBJ = 1u << 28, FL_APPLYABS = 1u << 29, FL_PROBE = 1u << 30, FL_ALIVE = 1u << 31;
// directDiffuse's thickness probe (frag.glsl:668): the ray starts ON the hit point (no 1e-4 offset) and traverses only the BVH
// of the object that was hit.  Probes exist in the RAYTRACING == 0 mode only, where the bounce counter and the index stack are
// not in use: while FL_PROBE is set their 16 bits (0-11 and 24-27) carry the object index.
constexpr int FL_PROBE_OBJ_MAX = 0xffff;
PM_DEV int probeObjOf(uint32_t f) { return (int)((f & FL_COUNT_MASK) | (((f >> FL_STACK_SHIFT) & 0xfu) << 12)); }

struct Path {
    vec3 O, D;             // current ray
    vec3 col, inc;         // throughput, incLight of the current sample
    vec3 sum;              // sum of samples of the current pixel-frame
    uint32_t rng;
    uint32_t pix, fi, ls;  // job identity: global pixel (x | y<<16), frame slot of the batch, accumulator slot
    int bounce, sample, stackSize;
    bool inObj, applyAbs, alive, probe; int probeObj;
    vec3 enter; float dist;   // RAY_ENTER_LOCATION, DISTANCE_TRAVELED
    uint32_t sc0, sc1, sc2; // refractionIndiceStack as dictionary codes, slot 0 in the lowest bits: 3-bit codes all in sc0; 8-bit codes: slots 0-3, 4-7, 8-9
    uint32_t sk[10];        // ... or, STK == 32 (a scene with more distinct Ni than the 8-bit dictionary holds), the ten floats themselves, as bits
    bool incNZ;            // FL_INCNZ
    bool g5loaded, g5dirty; // lazily fetched / modified (enter, dist) group, see k_shade
};

PM_DEV uint32_t packFlags(const Path& p) {
    const uint32_t lo = p.probe ? ((uint32_t)p.probeObj & FL_COUNT_MASK) : (uint32_t)p.bounce, st = p.probe ? ((uint32_t)p.probeObj >> 12) : (uint32_t)p.stackSize;
    return lo | ((uint32_t)p.sample << FL_SAMPLE_SHIFT) | (p.incNZ ? FL_INCNZ : 0u) | (st << FL_STACK_SHIFT) | (p.inObj ? FL_INOBJ : 0u) |
           (p.applyAbs ? FL_APPLYABS : 0u) | (p.alive ? FL_ALIVE : 0u) | (p.probe ? FL_PROBE : 0u);
}
PM_DEV void unpackFlags(Path& p, uint32_t f) {
    p.bounce = f & FL_COUNT_MASK; p.sample = (f >> FL_SAMPLE_SHIFT) & FL_SAMPLE_MASK; p.stackSize = (f >> FL_STACK_SHIFT) & 0xf; p.incNZ = f & FL_INCNZ;
    p.inObj = f & FL_INOBJ; p.applyAbs = f & FL_APPLYABS; p.alive = f & FL_ALIVE;
    p.probe = f & FL_PROBE; p.probeObj = p.probe ? probeObjOf(f) : 0;
}

// index stack, frag.glsl:139-158.  The shader's array shifts move slots [0, size] up (addToIndiceStack, only when size < 10) or slots [1, size) down
// (removeFirstOfIndiceStack) and leave every other slot as it was — stale values stay readable, and are read (:835, :839 with size 1).  On the packed
// codes that is one shift of the whole word merged under a mask of the slots the loop touches.  STK: 3 or 8 bits per code; 32: no codes, the floats' bits in ten registers (any number of distinct refraction indices).
// what a material pushes (:834) / what trace()'s prologue pushes (:816): its dictionary code, or the float's bits
template <int STK> PM_DEV uint32_t stackElemOf(const MatRec& m) { return STK == 32 ? __float_as_uint(m.Ni) : (uint32_t)m.niCode; }
template <int STK> PM_DEV uint32_t stackElemAir() { return STK == 32 ? __float_as_uint(1.0029f) : 1u; }
template <int STK> PM_DEV void addToIndiceStack(Path& p, uint32_t e) {
    if (p.stackSize < 10) {
        const int n = p.stackSize + 1;                           // slots 0 .. size are rewritten
        if (STK == 32) {
#pragma unroll
            for (int k = 9; k >= 1; k--) p.sk[k] = (k < n) ? p.sk[k - 1] : p.sk[k];
            p.sk[0] = e;
        } else if (STK == 3) {
            const uint32_t m = n >= 10 ? 0x3fffffffu : ((1u << (3 * n)) - 1u);
            p.sc0 = ((((p.sc0 << 3) | e) & m) | (p.sc0 & ~m));
        } else {
            const unsigned long long lo = (unsigned long long)p.sc0 | ((unsigned long long)p.sc1 << 32);
            const uint32_t hi = p.sc2;
            const unsigned long long slo = (lo << 8) | e; const uint32_t shi = ((hi << 8) | (uint32_t)(lo >> 56)) & 0xffffu;
            const unsigned long long mlo = n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1ull); const uint32_t mhi = n <= 8 ? 0u : ((1u << (8 * (n - 8))) - 1u);
            const unsigned long long nlo = (slo & mlo) | (lo & ~mlo);
            p.sc0 = (uint32_t)nlo; p.sc1 = (uint32_t)(nlo >> 32); p.sc2 = (shi & mhi) | (hi & ~mhi);
        }
        p.stackSize++;
    }
}
template <int STK> PM_DEV void removeFirstOfIndiceStack(Path& p) {
    if (p.stackSize > 0) {
        const int n = p.stackSize - 1;                           // slots 0 .. size-2 are rewritten
        if (STK == 32) {
#pragma unroll
            for (int k = 0; k < 9; k++) p.sk[k] = (k < n) ? p.sk[k + 1] : p.sk[k];
        } else if (STK == 3) {
            const uint32_t m = (1u << (3 * n)) - 1u;
            p.sc0 = ((p.sc0 >> 3) & m) | (p.sc0 & ~m);
        } else {
            const unsigned long long lo = (unsigned long long)p.sc0 | ((unsigned long long)p.sc1 << 32);
            const uint32_t hi = p.sc2;
            const unsigned long long slo = (lo >> 8) | ((unsigned long long)(hi & 0xffu) << 56); const uint32_t shi = hi >> 8;
            const unsigned long long mlo = n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1ull); const uint32_t mhi = n <= 8 ? 0u : ((1u << (8 * (n - 8))) - 1u);
            const unsigned long long nlo = (slo & mlo) | (lo & ~mlo);
            p.sc0 = (uint32_t)nlo; p.sc1 = (uint32_t)(nlo >> 32); p.sc2 = (shi & mhi) | (hi & ~mhi);
        }
        p.stackSize--;
    }
}
// refractionIndiceStack[slot] for slot 0 or 1 (the only slots trace() reads, :835-839)
template <int STK> PM_DEV float indiceStackSlot(const DevScene& sc, const Path& p, int slot) {
    if (STK == 32) return __uint_as_float(p.sk[slot]);
    if (STK == 3) {
        const uint32_t c = (p.sc0 >> (3 * slot)) & 7u;
        float v = sc.ni8[0];
#pragma unroll
        for (uint32_t k = 1; k < 8; k++) v = (c == k) ? sc.ni8[k] : v;           // seven selects on scalar operands, no memory
        return v;
    }
    return sc.niTable[(p.sc0 >> (8 * slot)) & 0xffu];
}

// texture(textures[0], uv): GL 4.6 §8.14 LINEAR/REPEAT on RGBA8 (dispatch.java:349-354)
PM_DEV int imod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
// byte / 255.0f, the UNORM8 -> float conversion of a GL_RGBA8 texel, correctly rounded without a division: one Newton step on the product with the
// rounded reciprocal.  Bit-equal to the IEEE quotient for all 256 bytes (tests/test_oracle_kat.py::test_unorm8_reciprocal_form, and on the device
// pt_debug_math fn 9); the oracle divides.
PM_DEV float unorm8(uint32_t b) {
    const float fb = (float)b, r = 0x1.010102p-8f;               // RN(1 / 255)
    const float q = fb * r;
    return fmaf(fmaf(-255.0f, q, fb), r, q);
}
PM_DEV float4 texel(const uchar4* tex, int idx) {
    const uchar4 t = tex[idx];
    return make_float4(unorm8(t.x), unorm8(t.y), unorm8(t.z), unorm8(t.w));
}
PM_DEV vec3 sampleTex(const uchar4* tex, int w, int h, float u, float v) {
    float fu = u * (float)w - 0.5f, fv = v * (float)h - 0.5f;
    float flu = (__builtin_fabsf(fu) < 1.0e9f) ? __builtin_floorf(fu) : 0.0f;
    float flv = (__builtin_fabsf(fv) < 1.0e9f) ? __builtin_floorf(fv) : 0.0f;
    float a = fu - flu, b = fv - flv;
    int i0 = imod((int)flu, w), j0 = imod((int)flv, h);
    int i1 = imod(i0 + 1, w), j1 = imod(j0 + 1, h);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 p00 = texel(tex, j0 * w + i0), p10 = texel(tex, j0 * w + i1), p01 = texel(tex, j1 * w + i0), p11 = texel(tex, j1 * w + i1);
    vec3 r;
    r.x = w00 * p00.x + w10 * p10.x + w01 * p01.x + w11 * p11.x;
    r.y = w00 * p00.y + w10 * p10.y + w01 * p01.y + w11 * p11.y;
    r.z = w00 * p00.z + w10 * p10.z + w01 * p01.z + w11 * p11.z;
    return r;
}
PM_DEV vec3 sampleSky(const DevScene& sc, float u, float v) { return sampleTex(sc.sky, sc.skyW, sc.skyH, u, v); }
PM_DEV vec3 sampleTexture(const DevScene& sc, int index, float u, float v) {     // frag.glsl:79-81
    const TexRec t = sc.tex[index];
    return sampleTex(t.data, t.w, t.h, u, v);
}
// uv of a triangle hit (frag.glsl:508-517) from the shading record; (-1,-1) when the triangle has no vt (Q-7)
PM_DEV void hitUV(const float4* S, float hu, float hv, float& uvx, float& uvy) {
    float4 s1 = S[1], s2 = S[2], s3 = S[3];
    float vt1x = s1.z, vt1y = s1.w;
    if (vt1x != 69.420f) {
        float w = 1.0f - hu - hv;
        uvx = s2.x * hu + s2.z * hv + w * vt1x;
        uvy = s2.y * hu + s3.x * hv + w * vt1y;
        uvy = 1.0f - uvy;
    } else { uvx = -1.0f; uvy = -1.0f; }
}
// hit.uvSample of any hit: a triangle's own uv; an ellipsoid's is the uv of the closest triangle the BVH loop found before it
// (hitUV is only ever written at :574), vec2(0) when there was none (:559)
PM_DEV void uvOfHit(const DevScene& sc, int prim, float hu, float hv, const float4* HX, unsigned slot, float& uvx, float& uvy) {
    if (prim & PRIM_ELLIPSOID) {
        const float4 x = HX[slot];
        const int tp = __float_as_int(x.z);
        if (tp >= 0) hitUV(sc.shade + 4 * (size_t)tp, x.x, x.y, uvx, uvy);
        else { uvx = 0.0f; uvy = 0.0f; }
    } else hitUV(sc.shade + 4 * (size_t)prim, hu, hv, uvx, uvy);
}
// mapMtl (frag.glsl:210-225) on the fields the render path reads, and the raw-texel normal of :827
PM_DEV void applyMaps(const DevScene& sc, MatRec& m, float u, float v, vec3& N) {
    if (m.map_Ka > -1) { vec3 t = sampleTexture(sc, m.map_Ka, u, v); m.Ka[0] = t.x * m.Ka[0]; m.Ka[1] = t.y * m.Ka[1]; m.Ka[2] = t.z * m.Ka[2]; }
    if (m.map_Kd > -1) { vec3 t = sampleTexture(sc, m.map_Kd, u, v); m.Kd[0] = t.x * m.Kd[0]; m.Kd[1] = t.y * m.Kd[1]; m.Kd[2] = t.z * m.Kd[2]; }
    if (m.map_Ks > -1) { vec3 t = sampleTexture(sc, m.map_Ks, u, v); m.Ks[0] = t.x; m.Ks[1] = t.y; m.Ks[2] = t.z; }
    if (m.map_Ke > -1) { vec3 t = sampleTexture(sc, m.map_Ke, u, v); m.Ke[0] = t.x; m.Ke[1] = t.y; m.Ke[2] = t.z; }
    if (m.map_Tr > -1) m.Tr = sampleTexture(sc, m.map_Tr, u, v).x;
    if (m.map_Pm > -1) m.Pm = sampleTexture(sc, m.map_Pm, u, v).x;
    if (m.map_Pr > -1) m.Pr = sampleTexture(sc, m.map_Pr, u, v).x;
    if (m.map_Pc > -1) m.Pc = sampleTexture(sc, m.map_Pc, u, v).x;
    if (m.map_norm > -1) N = sampleTexture(sc, m.map_norm, u, v);
}
PM_DEV vec3 bgCol(const DevScene& sc, vec3 In) {
    float u = 0.5f + atan2_(In.z, In.x) / (2.0f * 3.14159f);
    float v = 0.5f - asin_(In.y) / 3.14159f;
    return sampleSky(sc, u, v);
}

template <bool FAST = false>
PM_DEV float fresnelReflectAmount(float n1, float n2, vec3 normal, vec3 incidence) {
    float r0 = divT<FAST>(n1 - n2, n1 + n2);
    r0 *= r0;
    float cosX = -dot(normal, incidence);
    if (n1 > n2) {
        float n = divT<FAST>(n1, n2);
        float sinT2 = n * n * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = sqrtT<FAST>(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    return r0 + (1.0f - r0) * x * x * x * x * x;
}

// chooseRay, frag.glsl:745-809, split at its random-vector draw so that the kernel has ONE call site of
// randLambertianDistVec for all lobes (each inlined copy is a separate divergent instruction stream; the
// shading kernel is VALU-issue bound).  chooseLobe = weights + roll (+ the subsurface draw) -> winType;
// lobeDirection = the out direction for that winType from the already drawn Gaussian vector G.
// RNG draw order is the reference's: roll, [subsurface draw], 6 Gaussian draws (none for transmission).
template <bool FAST = false>
PM_DEV int chooseLobe(const MatRec& m, float n1, float n2, vec3 N, vec3 D, uint32_t& rng) {
    float reflectionWeight = 1.0f - m.Pr;
    float clearcoatWeight = m.Pc;
    float transmissionWeight = (m.Tr > 0.0f ? m.Tr : (m.Tf[0] > 0.0f ? (m.Tf[0] + m.Tf[1] + m.Tf[2]) / 3.0f : 0.0f));
    float subsurfaceWeight = m.subsurface;
    float fresnel = 0.0f;
    if (m.illum == 5 || m.illum == 7 || transmissionWeight > 0.0f) {
        fresnel = fresnelReflectAmount<FAST>(n1, n2, N, D);
        reflectionWeight += fresnel * m.Pr;
        transmissionWeight *= (1.0f - fresnel);
    }
    float diffuseWeight = (1.0f - m.Pm) * (1.0f - transmissionWeight) * (1.0f - fresnel);
    float totalWeight = diffuseWeight + reflectionWeight + clearcoatWeight + transmissionWeight;
    if (FAST) { const float inv = __builtin_amdgcn_rcpf(totalWeight); reflectionWeight *= inv; clearcoatWeight *= inv; transmissionWeight *= inv; }
    else { reflectionWeight /= totalWeight; clearcoatWeight /= totalWeight; transmissionWeight /= totalWeight; }
    float roll = random_(rng);
    if (roll < reflectionWeight) return 1;
    if (roll < reflectionWeight + clearcoatWeight) return 2;
    if (roll < reflectionWeight + clearcoatWeight + transmissionWeight) return 3;
    if (subsurfaceWeight > 0.0f) { if (random_(rng) < subsurfaceWeight) return 4; }
    return 0;
}
template <bool FAST = false>
PM_DEV vec3 lobeDirection(int w, vec3 G, vec3 N, vec3 D, float eta, float Pcr) {
    if (w == 3) return refractT<FAST>(D, N, eta);                                 // :783
    vec3 rough = normalizeT<FAST>(G + N);
    if (w == 1 || w == 2) return mix(reflect(D, N), rough, w == 1 ? 0.0f : Pcr);  // :775 (Q-8), :779
    return rough;                                                                 // :795-804
}

// Camera ray of one sample, main() frag.glsl:894-908, for global pixel (px,py): lens jitter (6 RNG draws), focus, normalise.
// (G = the Gaussian vector of the lens jitter, drawn by the caller: the shading kernel has ONE site of randLambertianDistVec for lobes and lenses)
template <bool FAST = false>
PM_DEV void cameraRayFrom(const FrameConst& fc, int W, int H, int px, int py, vec3 G, vec3& O, vec3& D) {
    float tcx = ((float)px + 0.5f) / (float)W, tcy = ((float)py + 0.5f) / (float)H;
    vec3 q = v3(((tcx * 2.0f - 1.0f) * -1.0f) * fc.screenSize, ((tcy * 2.0f - 1.0f) * fc.screenHratio) * fc.screenSize, fc.focalLength);
    vec3 direction = vecmat(q, fc.camRot);
    vec3 ORIGIN = v3(fc.origin[0], fc.origin[1], fc.origin[2]);
    vec3 origin_jittered = ORIGIN + vecmat(G * fc.BLUR, fc.camRot);
    vec3 focal_point = ORIGIN + direction * fc.focus;
    D = normalizeT<FAST>(focal_point - origin_jittered);
    O = origin_jittered;
}
template <bool FAST = false>
PM_DEV void cameraRay(const FrameConst& fc, int W, int H, int px, int py, uint32_t& rng, vec3& O, vec3& D) {
    cameraRayFrom<FAST>(fc, W, H, px, py, randLambertianDistVec<FAST>(rng), O, D);
}
// trace() prologue (:811-818): everything a new sample resets that needs no random numbers
template <int STK> PM_DEV void tracePrologue(Path& p) {
    p.col = v3(1.0f); p.inc = v3(0.0f);
    p.stackSize = 0;                       // clearIndiceStack
    if (STK) addToIndiceStack<STK>(p, stackElemAir<STK>()); // 1.0029 (dictionary code 1, or its bits); a scene without transmissive materials never reads the stack (:753)
    else p.stackSize = 1;
    p.inObj = false;
    p.bounce = 0;
    p.probe = false; p.probeObj = 0;
}
template <int STK, bool FAST = false>
PM_DEV void startSample(const FrameConst& fc, int W, int H, int px, int py, Path& p) {
    cameraRay<FAST>(fc, W, H, px, py, p.rng, p.O, p.D);
    tracePrologue<STK>(p);
}

// rngState = index + u_seed (:886,:896) for global pixel (px,py); false when the fragment returns early (:887)
PM_DEV bool pixelIndex(const FrameConst& fc, int W, int H, int px, int py, uint32_t& index) {
    float tcx = ((float)px + 0.5f) / (float)W, tcy = ((float)py + 0.5f) / (float)H;
    float resY = fc.resolution * fc.screenHratio;
    int pcx = (int)(tcx * fc.resolution), pcy = (int)(tcy * resY);
    index = (uint32_t)pcy * (uint32_t)fc.resolution + (uint32_t)pcx;
    return !(pcx >= (int)fc.resolution || pcy >= (int)resY);
}
PM_DEV bool inMouseOverlay(const FrameConst& fc, int px, int py) {          // :888, FRAME is left untouched there
    return __builtin_fabsf((float)px - fc.mouse[0]) < fc.resolution * 0.005f && __builtin_fabsf((float)py - fc.mouse[1]) < fc.resolution * 0.005f;
}

// Whether this iteration of trace()'s loop ends the sample (miss :876-878, cut-off :866, bounce budget :820) follows from the path state and the hit record alone —
// the shading kernel asks before it shades, so that its job pull and the loads only a finished sample needs are under way while it does.
PM_DEV bool segmentHit(float ht, int prim) { return !(prim == PRIM_NONE || !(ht < 1e25f)); }     // hit.id > -1 (:823) / closest_t < 1e25 (:634)
template <bool FAST = false>
PM_DEV bool segmentEndsSample(const FrameConst& fc, const Path& p, float ht, int prim) {
    return !segmentHit(ht, prim) || lengthT<FAST>(p.col) < 0.1f || !((float)(p.bounce + 1) < fc.MAX_BOUNCES);
}
// The lobe of a surface hit whose out direction still waits for its Gaussian vector (lobeDirection): the vector is drawn at the kernel's one site of
// randLambertianDistVec, which the lens jitter of new samples shares
struct LobePending { vec3 N; float Pcr; int w; bool needG; };

// One iteration of trace()'s while loop AFTER rayScene returned (frag.glsl:823-879), up to the Gaussian draw of chooseRay: the transmission lobe's direction is
// set here, the others' (L.needG) by the caller — p.D = lobeDirection(L.w, G, L.N, p.D, ., L.Pcr) with G = randLambertianDistVec(p.rng) — or, when the sample
// ends here, not at all: the six draws then only advance the stream (skipSixDraws).
// Returns true when the sample is finished (miss, cut-off, or bounce budget used up): segmentEndsSample's answer.
template <int STK, bool TEX, bool FAST = false>
PM_DEV bool shadeSegment(const DevScene& sc, const FrameConst& fc, Path& p, float ht, float hu, float hv, int prim, const float4* G5, const float4* HX, unsigned slot, LobePending& L) {
    constexpr bool TRANS = STK != 0;
    p.bounce++;                                               // :821
    const bool hit = segmentHit(ht, prim);
    const vec3 D = p.D;
    vec3 N = v3(0.0f), Ke = v3(0.0f), albedoKd = v3(0.0f), albedoKs = v3(0.0f), Tf = v3(0.0f);
    float ND = 0.0f, n1 = 1.0f, n2 = 1.0f, Pcr = 0.0f, Density = 0.0f;
    int w = 0;
    if (hit) {
        vec3 o = madd(D, 1e-4f, p.O);
        vec3 loc = madd(D, ht, o);                            // result.loc = o + closest_t*d (:635)
        int mat;
        if (prim & PRIM_ELLIPSOID) {
            const EllipRec& E = sc.ellip[prim & 0xffffff];
            vec3 c = v3(E.c[0], E.c[1], E.c[2]);
            if (E.rotated) N = normalizeT<FAST>(vecmat(loc - c, E.RB)); else N = normalizeT<FAST>(loc - c);     // :622-626
            mat = E.mat;
        } else {
            const float4* S = sc.shade + 4 * (size_t)prim;
            float4 s0 = S[0], s1 = S[1], s2 = S[2];
            vec3 vn1 = v3(s0.x, s0.y, s0.z), vn2 = v3(s0.w, s1.x, s1.y);
            if (vn1.x != 0.0f && vn1.y != 0.0f && vn1.z != 0.0f) N = normalizeT<FAST>(vn2 * hu + vn2 * hv + vn1 * (1.0f - hu - hv));   // :501-504 (Q-3)
            else N = vn2;                                                                                              // :506 (Q-4)
            mat = __float_as_int(s2.w);
        }
        MatRec m = sc.mats[mat];
        if (TEX && m.hasMaps) {                               // mapMtl + map_norm (:826-827)
            float uvx, uvy;
            uvOfHit(sc, prim, hu, hv, HX, slot, uvx, uvy);
            applyMaps(sc, m, uvx, uvy, N);
        }
        p.O = loc;                                            // :824
        ND = dot(N, D);
        N = N * (ND > 0.0f ? -1.0f : 1.0f);                   // :830
        if (TRANS) {
            const float s0 = indiceStackSlot<STK>(sc, p, 0), s1 = indiceStackSlot<STK>(sc, p, 1);
            // :833-836: slot 1 after the push is the old slot 0 — unless the stack was empty (the shift loop does not run: slot 1 keeps its stale
            // value) or full (the push is dropped: both slots stay)
            if (ND < 0.0f) { const bool full = p.stackSize >= 10; n1 = (full || p.stackSize == 0) ? s1 : s0; n2 = full ? s0 : m.Ni; addToIndiceStack<STK>(p, stackElemOf<STK>(m)); }
            else { n1 = s0; n2 = s1; removeFirstOfIndiceStack<STK>(p); }                       // :838-840
        }
        w = chooseLobe<FAST>(m, n1, n2, N, D, p.rng);         // :843 up to the lobe decision
        Ke = v3(m.Ke[0], m.Ke[1], m.Ke[2]); albedoKd = v3(m.Kd[0], m.Kd[1], m.Kd[2]); albedoKs = v3(m.Ks[0], m.Ks[1], m.Ks[2]);
        Tf = v3(m.Tf[0], m.Tf[1], m.Tf[2]); Pcr = m.Pcr; Density = m.Density;
    }
    L.N = N; L.Pcr = Pcr; L.w = w; L.needG = hit && w != 3;
    if (!hit) {
        p.inc = p.inc + bgCol(sc, D) * p.col;                 // :877
        return true;
    }
    if (w == 3) p.D = lobeDirection<FAST>(3, v3(0.0f), N, D, divT<FAST>(n1, n2), Pcr);      // :783: no random vector
    if (TRANS && w == 3) {                                    // :847-863
        if (!p.g5loaded) { float4 g5 = G5[slot]; p.enter = v3(g5.x, g5.y, g5.z); p.dist = g5.w; p.g5loaded = true; }   // only transmission touches it
        p.g5dirty = true;
        if (ND < 0.0f) {
            if (p.inObj) { p.dist = lengthT<FAST>(p.enter - p.O); p.applyAbs = true; }
            p.inObj = true;
            p.enter = p.O;
        } else {
            p.inObj = false;
            p.dist = lengthT<FAST>(p.enter - p.O);
            p.applyAbs = true;
        }
    }
    p.inc = p.inc + Ke * p.col;                               // :865
    if (lengthT<FAST>(p.col) < 0.1f) return true;             // :866
    if (TRANS && p.applyAbs) {
        p.col = p.col * exp3((-Tf) * p.dist * Density);      // :868
        p.applyAbs = false;
    } else if (w == 4) {
    } else {
        p.col = p.col * (w == 2 ? albedoKs : albedoKd);      // :873
    }
    return !((float)p.bounce < fc.MAX_BOUNCES);               // loop condition :820
}

// directDiffuse (frag.glsl:655-681), the RAYTRACING == 0 mode: ONE rayScene per sample, fixed up-light shading, and for
// subsurface > 0 a thickness probe = a second launch of this lane's ray through the hit object's BVH only.  The sample's
// radiance is left in p.inc; returns true when the sample is finished.  While a probe is in flight p.col carries
// directDiffuse's `o` (the camera ray origin) and p.inc.x the material index (both groups are always written back).
PM_DEV vec3 subsurfaceTint(const MatRec& m, vec3 o, vec3 loc) {
    float si = distance(o, loc);                                                                          // :668
    vec3 rad = v3(maxnum(m.ssRadius[0], 1e-4f), maxnum(m.ssRadius[1], 1e-4f), maxnum(m.ssRadius[2], 1e-4f));
    vec3 sigma_t = v3(1.0f / rad.x, 1.0f / rad.y, 1.0f / rad.z);                                          // :671
    return exp3((-sigma_t) * si) * v3(m.ssColor[0], m.ssColor[1], m.ssColor[2]);                          // :672
}
template <bool TEX>
PM_DEV bool directSegment(const DevScene& sc, Path& p, float ht, float hu, float hv, int prim, const float4* HX, unsigned slot) {
    if (p.probe) {                                            // second half of a subsurface sample: .loc of rayBVH is its (t,u,v) triple (:493)
        const MatRec m = sc.mats[__float_as_int(p.inc.x)];
        vec3 loc = (prim != PRIM_NONE) ? v3(ht, hu, hv) : v3(1e30f);
        p.inc = subsurfaceTint(m, p.col, loc);
        p.probe = false;
        return true;
    }
    const bool hit = !(prim == PRIM_NONE || !(ht < 1e25f));
    const vec3 D = p.D;
    if (!hit) { p.inc = bgCol(sc, D); return true; }          // :679
    vec3 o = madd(D, 1e-4f, p.O);
    vec3 loc = madd(D, ht, o);
    vec3 N; int mat;
    const bool ellipsoid = (prim & PRIM_ELLIPSOID) != 0;
    if (ellipsoid) {
        const EllipRec& E = sc.ellip[prim & 0xffffff];
        vec3 c = v3(E.c[0], E.c[1], E.c[2]);
        if (E.rotated) N = normalize(vecmat(loc - c, E.RB)); else N = normalize(loc - c);
        mat = E.mat;
    } else {
        const float4* S = sc.shade + 4 * (size_t)prim;
        float4 s0 = S[0], s1 = S[1], s2 = S[2];
        vec3 vn1 = v3(s0.x, s0.y, s0.z), vn2 = v3(s0.w, s1.x, s1.y);
        if (vn1.x != 0.0f && vn1.y != 0.0f && vn1.z != 0.0f) N = normalize(vn2 * hu + vn2 * hv + vn1 * (1.0f - hu - hv));
        else N = vn2;
        mat = __float_as_int(s2.w);
    }
    MatRec m = sc.mats[mat];
    if (TEX && m.hasMaps) {
        float uvx, uvy;
        uvOfHit(sc, prim, hu, hv, HX, slot, uvx, uvy);
        applyMaps(sc, m, uvx, uvy, N);
    }
    vec3 Kd = v3(m.Kd[0], m.Kd[1], m.Kd[2]);
    vec3 col = v3(m.Ka[0], m.Ka[1], m.Ka[2]) + Kd * 0.2f + (Kd * dot(v3(0.0f, 1.0f, 0.0f), N)) + v3(m.Ke[0], m.Ke[1], m.Ke[2]);   // :661 (N not flipped)
    if (m.subsurface > 0.0f) {
        // hit.parentID (:573) is the BVH of the closest TRIANGLE found, also when an ellipsoid in front of it won (:619-630 do not reset it)
        int triPrim = ellipsoid ? __float_as_int(hu) : prim;
        int obj = (triPrim >= 0) ? sc.triObj[triPrim] : -1;
        if (obj < 0) { p.inc = subsurfaceTint(m, p.O, v3(1e30f)); return true; }      // ellipsoid: parentID = -1, probe treated as a miss
        p.col = p.O;                                          // directDiffuse's `o`
        p.inc = v3(__int_as_float(mat), 0.0f, 0.0f);
        p.O = loc;                                            // the probe starts on the hit point, same direction
        p.probe = true; p.probeObj = obj;
        return false;
    }
    p.inc = col;
    return true;
}

}  // namespace ptd
