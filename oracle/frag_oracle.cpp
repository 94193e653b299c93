// oracle/frag_oracle.cpp — TEST INFRASTRUCTURE (parity oracle). Not part of the product.
//
// CPU float32 restatement of the reference's per-pixel Monte-Carlo render loop, i.e. of the GLSL
// fragment shader /root/reference/src/shaders/frag.glsl (main :884-934, trace :810-882,
// chooseRay :745-809, rayScene :548-653, rayBVH :452-537, rayTri :351-372, rayEllipsoid :373-384,
// rayBox :408-419, RNG :683-708, bgCol :235-242, rotate/rotateBack :244-297, newMtl :170-209,
// index stack :136-158), driven through the same buffers the reference binds as SSBOs
// (/root/reference/src/Main/dispatch.java:191-211, 270-329, 386-534, 554-574).
//
// PARITY PIN: the reference ships no tests, golden vectors or reproducible images and nothing
// of it can be executed in this environment (no JDK, no GL) -> "parity unpinned" by the reference
// itself.  This restatement is pinned by (1) the integer-exact RNG vectors of SURVEY.md §8(c),
// (2) analytic known-answer tests K2-K9 (tests/test_oracle_kat.py) whose answers follow from the
// cited shader lines, and (3) line-by-line review.  Quirks Q-1..Q-17 of SURVEY.md are kept.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "glsl_math.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <cstdio>
#include <thread>
#include <vector>

using namespace gm;

extern "C" {

struct orc_scene {
    const float* origin;        // binding 0, 3 f32
    const float* rotation;      // binding 1, 3 f32
    const float* mouse;         // binding 2, 3 f32
    const float* tris;          // binding 3, 40 f32 per triangle
    int64_t n_tris;
    const float* params;        // binding 4, 12 f32
    const float* imp;           // binding 5
    const float* ellip;         // binding 7
    const float* bvhdata;       // binding 10, 8 f32 per node
    const int32_t* bvhtree;     // binding 11, 3 i32 per node
    int64_t n_nodes;
    const int32_t* leaf_tris;   // binding 12
    int64_t n_leaf_tris;
    const int32_t* obj_indices; // binding 13, [count, roots...]
    const float* mtl;           // binding 14, [48.0, 48 f32 per material]
    int64_t n_mtl_floats;
    const uint8_t* sky;         // texture 0, RGBA8, row 0 first
    int32_t sky_w, sky_h;
    // binding 15: the bindless texture table (dispatch.java:334-378); entry 0 = sky.  RGBA8, row 0 first; NULL = not uploaded
    const uint8_t* const* tex;
    const int32_t* tex_w;
    const int32_t* tex_h;
    int32_t n_tex;
};

// counters: S, Nv, Tt, Hu + extras
// ... and the shape of the traversal (what a kernel that takes two steps per trip could chain): inner nodes popped; those popped right after their
// parent's step as its left / its right child (the near child, pushed last); leaves popped right after their parent's step, and those of them that hold one triangle
// ... and (round 6) a WHAT-IF pass over the same rays (orc_set_whatif): nodes popped / triangles tested by another order of the object loop, and the rays whose
// hit record would then differ from the reference's (DESIGN.md "object order")
enum { C_SEGMENTS = 0, C_NODES, C_TRITESTS, C_HITUPD, C_SAMPLES, C_BOXTESTS, C_RNG, C_ELLIP, C_INNER, C_LEFTNEXT, C_RIGHTNEXT, C_LEAFNEXT, C_LEAFNEXT1,
       C_XNODES, C_XTRIS, C_XDIFF, C_XPRE, C_N };

}  // extern "C"

namespace {

struct Mtl {
    vec3 Ka, Kd, Ks; float Ns, d, Tr; vec3 Tf; float Ni; vec3 Ke; float Density; int illum;
    int map_Ka, map_Kd, map_Ks; float Pm, Pr, Ps, Pc, Pcr, aniso, anisor;
    int map_Pm, map_Pr, map_Ps, map_Pc, map_Pcr, map_norm, map_d, map_Tr, map_Ns, map_Ke;
    float subsurface; vec3 subsurfaceColor, subsurfaceRadius;
};

struct Hit {          // raySceneResult, frag.glsl:83-96 (only the members trace() consumes)
    vec3 loc, norm; int material; int id; int type; float distance; vec3 dir; float uvx, uvy; int parentID;
};

struct Mat3 { float m[3][3]; };   // math (row-major) matrix

struct Ctx {
    const orc_scene* s;
    // Parameters block (frag.glsl:39-52)
    float screenSize, focalLength, resolution, screenHratio, SAMPLE_RES, MAX_BOUNCES, GAMMA, BLUR,
        FOCAL_DISTANCE, RAYTRACING, DEBUG, AUTO_FOCUS;
    vec3 ORIGIN, ROTATION, MOUSE_POS;
    int me, numObj, numImplicits, numEllipsoids;
    Mat3 camRot;                  // rotationMatrix(ROTATION), uniform per frame
    uint64_t* cnt;                // per-thread counters
    bool count;                   // false while casting the auto-focus ray
};

// per-invocation shader "globals" (frag.glsl:136-166).  GLSL leaves uninitialised globals
// undefined; the oracle defines them as zero/false.
struct Inv {
    float stack[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int stackSize = 0;
    bool RAY_IN_OBJECT = false, APPLY_ABSORBTION = false;
    vec3 RAY_ENTER_LOCATION = {0, 0, 0};
    float DISTANCE_TRAVELED = 0;
};

// ---- index stack, frag.glsl:139-158 ----
void clearIndiceStack(Inv& g) { g.stackSize = 0; }
void addToIndiceStack(Inv& g, float e) {
    if (g.stackSize < 10) {
        for (int i = g.stackSize; i > 0; i--) g.stack[i] = g.stack[i - 1];
        g.stack[0] = e;
        g.stackSize++;
    }
}
void removeFirstOfIndiceStack(Inv& g) {
    if (g.stackSize > 0) {
        for (int i = 0; i < g.stackSize - 1; i++) g.stack[i] = g.stack[i + 1];
        g.stackSize--;
    }
}

// ---- materials, frag.glsl:170-209 ----
Mtl newMtl(const Ctx& c, int m) {
    const float* d = c.s->mtl; int me = c.me;
    auto F = [&](int k) { return d[me * m + k]; };
    Mtl o;
    o.Ka = v3(F(1), F(2), F(3)); o.Kd = v3(F(4), F(5), F(6)); o.Ks = v3(F(7), F(8), F(9));
    o.Ns = F(10); o.d = F(11); o.Tr = F(12); o.Tf = v3(F(13), F(14), F(15)); o.Ni = F(16);
    o.Ke = v3(F(17), F(18), F(19)); o.Density = F(20); o.illum = (int)F(21);
    o.map_Ka = (int)F(22); o.map_Kd = (int)F(23); o.map_Ks = (int)F(24);
    o.Pm = F(25); o.Pr = F(26); o.Ps = F(27); o.Pc = F(28); o.Pcr = F(29); o.aniso = F(30); o.anisor = F(31);
    o.map_Pm = (int)F(32); o.map_Pr = (int)F(33); o.map_Ps = (int)F(34); o.map_Pc = (int)F(35);
    o.map_Pcr = (int)F(36); o.map_norm = (int)F(37); o.map_d = (int)F(38); o.map_Tr = (int)F(39);
    o.map_Ns = (int)F(40); o.map_Ke = (int)F(41);
    o.subsurface = F(42); o.subsurfaceColor = v3(F(43), F(44), F(45)); o.subsurfaceRadius = v3(F(46), F(47), F(48));
    return o;
}

// ---- texture(): GL 4.6 §8.14, LINEAR min/mag, REPEAT wrap, RGBA8 UNORM (dispatch.java:349-354) ----
int imod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
vec3 sampleTex(const uint8_t* px, int w, int h, float u, float v) {
    float fu = u * (float)w - 0.5f, fv = v * (float)h - 0.5f;
    float flu = (__builtin_fabsf(fu) < 1.0e9f) ? __builtin_floorf(fu) : 0.0f;
    float flv = (__builtin_fabsf(fv) < 1.0e9f) ? __builtin_floorf(fv) : 0.0f;
    float a = fu - flu, b = fv - flv;
    int i0 = imod((int)flu, w), j0 = imod((int)flv, h);
    int i1 = imod(i0 + 1, w), j1 = imod(j0 + 1, h);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float out[3];
    for (int k = 0; k < 3; k++) {
        float t00 = (float)px[4 * (j0 * w + i0) + k] / 255.0f, t10 = (float)px[4 * (j0 * w + i1) + k] / 255.0f;
        float t01 = (float)px[4 * (j1 * w + i0) + k] / 255.0f, t11 = (float)px[4 * (j1 * w + i1) + k] / 255.0f;
        out[k] = w00 * t00 + w10 * t10 + w01 * t01 + w11 * t11;
    }
    return v3(out[0], out[1], out[2]);
}
vec3 sampleSky(const Ctx& c, float u, float v) { return sampleTex(c.s->sky, c.s->sky_w, c.s->sky_h, u, v); }
// sampleTexture(), frag.glsl:79-81
vec3 sampleTexture(const Ctx& c, int index, float u, float v) {
    const orc_scene* s = c.s;
    if (index == 0 || !s->tex) return sampleSky(c, u, v);
    return sampleTex(s->tex[index], s->tex_w[index], s->tex_h[index], u, v);
}
// mapMtl(), frag.glsl:210-225: mapped materials are reset to mapped values, otherwise unchanged
Mtl mapMtl(const Ctx& c, const Mtl& M, float u, float v) {
    Mtl m = M;
    if (M.map_Ka > -1) m.Ka = sampleTexture(c, M.map_Ka, u, v) * M.Ka;
    if (M.map_Kd > -1) m.Kd = sampleTexture(c, M.map_Kd, u, v) * M.Kd;
    if (M.map_Ks > -1) m.Ks = sampleTexture(c, M.map_Ks, u, v);
    if (M.map_Ke > -1) m.Ke = sampleTexture(c, M.map_Ke, u, v);
    if (M.map_d > -1) m.d = sampleTexture(c, M.map_d, u, v).x;
    if (M.map_Tr > -1) m.Tr = sampleTexture(c, M.map_Tr, u, v).x;
    if (M.map_Ns > -1) m.Ns = sampleTexture(c, M.map_Ns, u, v).x;
    if (M.map_Pm > -1) m.Pm = sampleTexture(c, M.map_Pm, u, v).x;
    if (M.map_Pr > -1) m.Pr = sampleTexture(c, M.map_Pr, u, v).x;
    if (M.map_Ps > -1) m.Ps = sampleTexture(c, M.map_Ps, u, v).x;
    if (M.map_Pc > -1) m.Pc = sampleTexture(c, M.map_Pc, u, v).x;
    return m;
}
// frag.glsl:235-242
vec3 bgCol(const Ctx& c, vec3 In) {
    float u = 0.5f + atan2_(In.z, In.x) / (2.0f * 3.14159f);
    float v = 0.5f - asin_(In.y) / 3.14159f;
    return sampleSky(c, u, v);
}

// ---- rotation, frag.glsl:244-297 ----
Mat3 matmul(const Mat3& A, const Mat3& B) {
    Mat3 R;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return R;
}
// rotationMatrix(angles) = rotateX * rotateY * (z != 0 ? rotateZ : I); GLSL constructors are
// column-major, the Mat3 here is the same matrix written row-major.
Mat3 rotationMatrix(vec3 a) {
    float cx = cos_(a.x), sx = sin_(a.x), cy = cos_(a.y), sy = sin_(a.y);
    Mat3 RX = {{{1, 0, 0}, {0, cx, sx}, {0, -sx, cx}}};
    Mat3 RY = {{{cy, 0, -sy}, {0, 1, 0}, {sy, 0, cy}}};
    Mat3 RZ = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    if (a.z != 0.0f) { float cz = cos_(a.z), sz = sin_(a.z); RZ = {{{cz, sz, 0}, {-sz, cz, 0}, {0, 0, 1}}}; }
    return matmul(matmul(RX, RY), RZ);
}
// p * rm  (row vector times matrix): component j = dot(p, column j)
vec3 vecmat(vec3 p, const Mat3& M) {
    return v3(dot(p, v3(M.m[0][0], M.m[1][0], M.m[2][0])), dot(p, v3(M.m[0][1], M.m[1][1], M.m[2][1])),
              dot(p, v3(M.m[0][2], M.m[1][2], M.m[2][2])));
}
vec3 rotate(vec3 p, vec3 rot) { return vecmat(p, rotationMatrix(rot)); }
Mat3 rotateBackMatrix(vec3 rot) {
    float cx = cos_(rot.x), sx = sin_(rot.x), cy = cos_(rot.y), sy = sin_(rot.y), cz = cos_(rot.z), sz = sin_(rot.z);
    // columns as written at frag.glsl:291-295
    float c0[3] = {cy * cz, cx * sz + cz * sx * sy, sx * sz - cx * cz * sy};
    float c1[3] = {-cy * sz, cx * cz - sx * sy * sz, cz * sx + cx * sy * sz};
    float c2[3] = {sy, -cy * sx, cx * cy};
    Mat3 M;
    for (int r = 0; r < 3; r++) { M.m[r][0] = c0[r]; M.m[r][1] = c1[r]; M.m[r][2] = c2[r]; }
    return M;
}
vec3 rotateBack(vec3 p, vec3 rot) { return vecmat(p, rotateBackMatrix(rot)); }

// ---- intersection primitives ----
// frag.glsl:351-372
bool rayTri(vec3 o, vec3 d, vec3 v1, vec3 v2, vec3 v3_, float& t, float& u, float& v) {
    const float EPSILON = 1e-10f;
    vec3 e1 = v2 - v1, e2 = v3_ - v1;
    vec3 dCross_e2 = cross(d, e2);
    float det = dot(e1, dCross_e2);
    if (__builtin_fabsf(det) < EPSILON) return false;
    float invDet = 1.0f / det;
    vec3 s = o - v1;
    u = dot(s, dCross_e2) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    vec3 sCross_e1 = cross(s, e1);
    v = dot(d, sCross_e1) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(e2, sCross_e1) * invDet;
    return t > EPSILON;
}
// frag.glsl:373-384 (note the operator precedence of the condition)
float rayEllipsoid(vec3 o, vec3 d, vec3 c, float r, float f, float g, float h) {
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
// frag.glsl:408-419
float rayBox(vec3 o, vec3 invD, const float* b) {
    float tminx = (b[0] - o.x) * invD.x, tminy = (b[1] - o.y) * invD.y, tminz = (b[2] - o.z) * invD.z;
    float tmaxx = (b[3] - o.x) * invD.x, tmaxy = (b[4] - o.y) * invD.y, tmaxz = (b[5] - o.z) * invD.z;
    float t1x = minnum(tminx, tmaxx), t1y = minnum(tminy, tmaxy), t1z = minnum(tminz, tmaxz);
    float t2x = maxnum(tminx, tmaxx), t2y = maxnum(tminy, tmaxy), t2z = maxnum(tminz, tmaxz);
    float tNear = maxnum(maxnum(t1x, t1y), t1z);
    float tFar = minnum(minnum(t2x, t2y), t2z);
    return (tFar >= tNear && tFar > 0.0f) ? (tNear > 0.0f ? tNear : 0.0f) : 1e30f;
}

struct BvhResult { float t, u, v; int id; bool any; };

// frag.glsl:452-537.  Returns the closest hit below previous_closest_t, if any.
BvhResult rayBVH(Ctx& c, vec3 o, vec3 d, int top, float previous_closest_t) {
    const orc_scene* s = c.s;
    BvhResult res{1e30f, 0, 0, -1, false};
    float closest_t = previous_closest_t;
    vec3 invD = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int stack[64]; int sp = 0;
    if (c.count) c.cnt[C_BOXTESTS]++;
    if (rayBox(o, invD, s->bvhdata + 8 * top) > closest_t) return res;
    stack[sp++] = top;
    int prevInner = -1;                          // (statistics only) the inner node of the previous iteration, -1 after a leaf
    while (sp > 0) {
        int node = stack[--sp];
        if (c.count) c.cnt[C_NODES]++;
        int left = s->bvhtree[3 * node + 1], right = s->bvhtree[3 * node + 2];
        bool isLeaf = (left | right) == -1;
        if (c.count) {
            const bool fromParent = prevInner >= 0 && (node == s->bvhtree[3 * prevInner + 1] || node == s->bvhtree[3 * prevInner + 2]);
            if (!isLeaf) {
                c.cnt[C_INNER]++;
                if (fromParent) c.cnt[node == s->bvhtree[3 * prevInner + 1] ? C_LEFTNEXT : C_RIGHTNEXT]++;
            } else if (fromParent) {
                c.cnt[C_LEAFNEXT]++;
                if ((int)s->bvhdata[8 * node + 7] - (int)s->bvhdata[8 * node + 6] == 1) c.cnt[C_LEAFNEXT1]++;
            }
            prevInner = isLeaf ? -1 : node;
        }
        if (isLeaf) {
            int startIdx = (int)s->bvhdata[8 * node + 6], endIdx = (int)s->bvhdata[8 * node + 7];
            for (int i = startIdx; i < endIdx; i++) {
                int tri = s->leaf_tris[i];
                const float* T = s->tris + 40 * (int64_t)tri;
                float t = 0, u = 0, v = 0;
                if (c.count) c.cnt[C_TRITESTS]++;
                bool ok = rayTri(o, d, v3(T[0], T[1], T[2]), v3(T[4], T[5], T[6]), v3(T[8], T[9], T[10]), t, u, v);
                float hx = ok ? t : 1e30f;
                if (hx > 0.0f && hx < closest_t) {
                    closest_t = hx;
                    if (c.count) c.cnt[C_HITUPD]++;
                    res.t = hx; res.u = u; res.v = v; res.id = tri; res.any = true;
                }
            }
        } else {
            if (c.count) c.cnt[C_BOXTESTS] += 2;
            float Ld = rayBox(o, invD, s->bvhdata + 8 * (left > 0 ? left : 0));
            float Rd = rayBox(o, invD, s->bvhdata + 8 * (right > 0 ? right : 0));
            if (Ld > Rd) {
                if (Ld < closest_t) stack[sp++] = left;
                if (Rd < closest_t) stack[sp++] = right;
            } else {
                if (Rd < closest_t) stack[sp++] = right;
                if (Ld < closest_t) stack[sp++] = left;
            }
        }
    }
    return res;
}

// normal / uv selection for the winning triangle, frag.glsl:499-518.
void triShading(const Ctx& c, int tri, float u, float v, vec3& norm, float& uvx, float& uvy, int& mtl) {
    const float* T = c.s->tris + 40 * (int64_t)tri;
    mtl = (int)T[36];
    vec3 n1 = v3(T[12], T[13], T[14]);
    if (n1.x != 0.0f && n1.y != 0.0f && n1.z != 0.0f) {
        vec3 n2 = v3(T[16], T[17], T[18]);
        vec3 n3 = n2;                                 // Q-3: third vertex normal := second
        norm = normalize(n2 * u + n3 * v + n1 * (1.0f - u - v));
    } else {
        norm = v3(T[16], T[17], T[18]);               // Q-4: raw n2
    }
    float vt1x = T[24], vt1y = T[25];
    if (vt1x != 69.420f) {
        float w = 1.0f - u - v;
        uvx = T[28] * u + T[32] * v + w * vt1x;
        uvy = T[29] * u + T[33] * v + w * vt1y;
        uvy = 1.0f - uvy;
    } else { uvx = -1.0f; uvy = -1.0f; }
}


// ---- what-if statistics (round 6; never part of a rendered value) --------------------------------------------------------------------
// The object loop of rayScene (:563-577) visits the BVHs in index order.  These restate it in other orders ON THE SAME RAYS and count what a kernel built that
// way would visit, and how often its hit record would differ from the reference's.
//   1  nearest root box first; a BVH of lower index than the current winner's is entered with `<=` instead of `<` (bound = nextup(closest_t)) until it produces a hit
//   2  the same, and while `<=` holds the pushes are pruned against closest_t * (1 + m) (a margin for box distances that round above their triangles')
//   3  a bounding pre-pass through the BVH of the nearest root box, then the reference's loop with closest_t starting at U = t_pre * (1 + m) (no hit recorded)
//   4  pre-pass = mode 1 over all BVHs, then the reference's loop from U
//   5  pre-pass = BVHs nearest root first until one produces a hit, then the reference's loop from U
//   6  pre-pass = BVHs nearest root first, stopping at the FIRST triangle hit, then the reference's loop from U
//   7 / 8  mode 1 with only the one / two nearest root boxes taken out of the index order
int g_whatif = 0;
float g_whatif_margin = 1.0f / 64.0f;

struct XStat { uint64_t nodes = 0, tris = 0; };

// rayBVH with the acceptance / pruning rule as parameters.  closest: bound; eq: accept t <= closest (and prune by <=) until the first hit of this call; scale: factor on
// the pruning bound while eq holds; firstHit: return at the first accepted triangle
BvhResult xBVH(Ctx& c, vec3 o, vec3 d, int top, float closest, bool eq, float scale, bool firstHit, XStat& st) {
    const orc_scene* s = c.s;
    BvhResult res{1e30f, 0, 0, -1, false};
    vec3 invD = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int stack[64]; int sp = 0;
    auto pruneBound = [&]() { return eq ? __builtin_nextafterf(closest, __builtin_inff()) * scale : closest; };
    if (rayBox(o, invD, s->bvhdata + 8 * top) > (eq ? closest * scale : closest)) return res;
    stack[sp++] = top;
    while (sp > 0) {
        int node = stack[--sp];
        st.nodes++;
        int left = s->bvhtree[3 * node + 1], right = s->bvhtree[3 * node + 2];
        if ((left | right) == -1) {
            int startIdx = (int)s->bvhdata[8 * node + 6], endIdx = (int)s->bvhdata[8 * node + 7];
            for (int i = startIdx; i < endIdx; i++) {
                int tri = s->leaf_tris[i];
                const float* T = s->tris + 40 * (int64_t)tri;
                float t = 0, u = 0, v = 0;
                st.tris++;
                bool ok = rayTri(o, d, v3(T[0], T[1], T[2]), v3(T[4], T[5], T[6]), v3(T[8], T[9], T[10]), t, u, v);
                float hx = ok ? t : 1e30f;
                if (hx > 0.0f && (eq ? hx <= closest : hx < closest)) {
                    closest = hx; eq = false;
                    res.t = hx; res.u = u; res.v = v; res.id = tri; res.any = true;
                    if (firstHit) return res;
                }
            }
        } else {
            float Ld = rayBox(o, invD, s->bvhdata + 8 * (left > 0 ? left : 0));
            float Rd = rayBox(o, invD, s->bvhdata + 8 * (right > 0 ? right : 0));
            float pb = pruneBound();
            if (Ld > Rd) {
                if (Ld < pb) stack[sp++] = left;
                if (Rd < pb) stack[sp++] = right;
            } else {
                if (Rd < pb) stack[sp++] = right;
                if (Ld < pb) stack[sp++] = left;
            }
        }
    }
    return res;
}

struct XHit { float t; int id; int obj; };

// the BVH part of rayScene in the order `mode` names; o is the offset origin
XHit xObjectLoop(Ctx& c, vec3 o, vec3 d, int mode, XStat& st, XStat& pre) {
    const orc_scene* s = c.s;
    const int n = c.numObj;
    vec3 invD = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    std::vector<std::pair<float, int>> order;                      // (root entry distance, object), root boxes the ray meets only
    for (int I = 1; I < n + 1; I++) {
        float e = rayBox(o, invD, s->bvhdata + 8 * s->obj_indices[I]);
        if (!(e > 1e29f)) order.push_back({e, I});
    }
    std::stable_sort(order.begin(), order.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    const float m = g_whatif_margin;
    auto nearestFirst = [&](bool margin, bool untilHit, bool firstTri, XStat& S) {
        XHit h{1e30f, -1, 1 << 30};
        for (auto& e : order) {
            int I = e.second;
            bool eq = h.id >= 0 && I < h.obj;
            BvhResult r = xBVH(c, o, d, s->obj_indices[I], h.t, eq, (eq && margin) ? 1.0f + m : 1.0f, firstTri, S);
            if (r.any) { h.t = r.t; h.id = r.id; h.obj = I; if (untilHit || firstTri) break; }
        }
        return h;
    };
    auto indexOrderFrom = [&](float U) {
        XHit h{U, -1, -1};
        for (int I = 1; I < n + 1; I++) {
            BvhResult r = xBVH(c, o, d, s->obj_indices[I], h.t, false, 1.0f, false, st);
            if (r.any && r.t < h.t) { h.t = r.t; h.id = r.id; h.obj = I; }
        }
        if (h.id < 0) h.t = 1e30f;
        return h;
    };
    if (mode == 9) {                                                // 9: index order over the BVHs whose root box the ray meets (what the per-ray cull leaves today)
        std::sort(order.begin(), order.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.second < b.second; });
        return nearestFirst(false, false, false, st);
    }
    if (mode == 7 || mode == 8) {                                   // 7 / 8: the one / two nearest root boxes first, the rest in index order
        size_t k = std::min(order.size(), (size_t)(mode - 6));
        std::sort(order.begin() + k, order.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.second < b.second; });
        return nearestFirst(false, false, false, st);
    }
    if (mode == 1) return nearestFirst(false, false, false, st);
    if (mode == 2) return nearestFirst(true, false, false, st);
    XHit p{1e30f, -1, -1};
    if (mode == 3) {
        if (!order.empty()) {
            BvhResult r = xBVH(c, o, d, s->obj_indices[order[0].second], 1e30f, false, 1.0f, false, pre);
            if (r.any) { p.t = r.t; p.id = r.id; }
        }
    } else if (mode == 4) p = nearestFirst(false, false, false, pre);
    else if (mode == 5) p = nearestFirst(false, true, false, pre);
    else p = nearestFirst(false, false, true, pre);
    if (p.id < 0 && mode != 3) return XHit{1e30f, -1, -1};          // every BVH the ray meets traversed without a bound and nothing found: the reference finds nothing
    return indexOrderFrom(p.id >= 0 ? p.t * (1.0f + m) : 1e30f);
}

// frag.glsl:548-653
Hit rayScene(Ctx& c, vec3 o, vec3 d) {
    const orc_scene* s = c.s;
    o = madd(d, 1e-4f, o);
    vec3 N = v3(0.0f);
    float closest_t = 1e30f;
    int hitType = 0, hitID = -1, hitMat = -1;
    float uvx = 0, uvy = 0;
    float bu = 0, bv = 0; int btri = -1; int parentID = -1;
    for (int I = 1; I < c.numObj + 1; I++) {
        int root = s->obj_indices[I];
        BvhResult r = rayBVH(c, o, d, root, closest_t);
        if (r.any && r.t < closest_t) { closest_t = r.t; hitType = 1; hitID = r.id; btri = r.id; bu = r.u; bv = r.v; parentID = root; }
    }
    if (g_whatif && c.count) {
        XStat st, pre;
        XHit x = xObjectLoop(c, o, d, g_whatif, st, pre);
        c.cnt[C_XNODES] += st.nodes + pre.nodes; c.cnt[C_XTRIS] += st.tris + pre.tris; c.cnt[C_XPRE] += pre.nodes;
        float rt = btri >= 0 ? closest_t : 1e30f;
        if (x.id != btri || std::memcmp(&x.t, &rt, 4) != 0) c.cnt[C_XDIFF]++;
    }
    if (btri >= 0) triShading(c, btri, bu, bv, N, uvx, uvy, hitMat);
    // implicits (frag.glsl:578-605): rayImplicit returns 1e30 unconditionally (:385-386); the
    // test `t < closest_t` can then only pass when closest_t > 1e30, i.e. never: whatever
    // ImpData holds, the loop changes nothing and draws no random number.
    const float* E = s->ellip; int n = c.numEllipsoids;
    for (int i = 0; i < n; i++) {
        vec3 cc = v3(E[1 + 3 * i], E[1 + 3 * i + 1], E[1 + 3 * i + 2]);
        vec3 st = v3(E[1 + n * 3 + 3 * i], E[1 + n * 3 + 3 * i + 1], E[1 + n * 3 + 3 * i + 2]);
        vec3 rot = v3(E[1 + n * 6 + 3 * i], E[1 + n * 6 + 3 * i + 1], E[1 + n * 6 + 3 * i + 2]);
        float r = E[1 + n * 9 + i];
        int mat = (int)E[1 + n * 10 + i];
        bool rotated = length(rot) > 0.0f;
        float t;
        if (c.count) c.cnt[C_ELLIP]++;
        if (rotated) t = rayEllipsoid(rotate(o, rot), rotate(d, rot), cc, r, st.x, st.y, st.z);
        else t = rayEllipsoid(o, d, cc, r, st.x, st.y, st.z);
        if (t < closest_t) {
            closest_t = t; hitMat = mat;
            if (rotated) N = normalize(rotateBack(madd(d, t, o) - cc, rot));
            else N = normalize(madd(d, t, o) - cc);
            hitType = 3; hitID = i;
        }
    }
    Hit h;
    h.type = hitType; h.id = hitID; h.uvx = uvx; h.uvy = uvy; h.parentID = parentID;
    if (closest_t < 1e25f) {
        h.loc = madd(d, closest_t, o); h.dir = normalize(d); h.norm = N; h.material = hitMat; h.distance = closest_t;
        return h;
    }
    h.loc = v3(1e30f); h.dir = d; h.norm = v3(0.0f); h.material = -1; h.distance = -1.0f;
    return h;
}

// ---- RNG, frag.glsl:686-708 ----
uint32_t NextRandom(uint32_t& state) {
    state = state * 747796405u + 2891336453u;
    uint32_t result = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    result = (result >> 22u) ^ result;
    return result;
}
float random_(Ctx& c, uint32_t& state) {
    if (c.count) c.cnt[C_RNG]++;
    return (float)NextRandom(state) / 4294967295.0f;    // the literal rounds to 2^32 in binary32
}
float randValNormalDist(Ctx& c, uint32_t& st) {
    float theta = 2.0f * 3.1415926f * random_(c, st);
    float rho = __builtin_sqrtf(-2.0f * log_(random_(c, st)));
    return rho * cos_(theta);
}
vec3 randLambertianDistVec(Ctx& c, uint32_t& st) {
    float x = randValNormalDist(c, st), y = randValNormalDist(c, st), z = randValNormalDist(c, st);
    return v3(x, y, z);
}

// frag.glsl:726-743
float fresnelReflectAmount(float n1, float n2, vec3 normal, vec3 incidence) {
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -dot(normal, incidence);
    if (n1 > n2) {
        float n = n1 / n2;
        float sinT2 = n * n * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = __builtin_sqrtf(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    return r0 + (1.0f - r0) * x * x * x * x * x;
}

// frag.glsl:745-809
vec3 chooseRay(Ctx& c, const Mtl& m, float n1, float n2, vec3 N, vec3 D, uint32_t& rng, int& winType) {
    float reflectionWeight = 1.0f - m.Pr;
    float clearcoatWeight = m.Pc;
    float transmissionWeight = (m.Tr > 0.0f ? m.Tr : (m.Tf.x > 0.0f ? (m.Tf.x + m.Tf.y + m.Tf.z) / 3.0f : 0.0f));
    float subsurfaceWeight = m.subsurface;
    float eta = n1 / n2;
    float fresnel = 0.0f;
    if (m.illum == 5 || m.illum == 7 || transmissionWeight > 0.0f) {
        fresnel = fresnelReflectAmount(n1, n2, N, D);
        reflectionWeight += fresnel * m.Pr;
        transmissionWeight *= (1.0f - fresnel);
    }
    float diffuseWeight = (1.0f - m.Pm) * (1.0f - transmissionWeight) * (1.0f - fresnel);
    float totalWeight = diffuseWeight + reflectionWeight + clearcoatWeight + transmissionWeight;
    diffuseWeight /= totalWeight; reflectionWeight /= totalWeight; clearcoatWeight /= totalWeight; transmissionWeight /= totalWeight;
    (void)diffuseWeight;
    float roll = random_(c, rng);
    vec3 outDir;
    if (roll < reflectionWeight) {
        winType = 1;
        outDir = mix(reflect(D, N), normalize(randLambertianDistVec(c, rng) + N), 0.0f);   // Q-8: 6 draws, discarded
    } else if (roll < reflectionWeight + clearcoatWeight) {
        winType = 2;
        outDir = mix(reflect(D, N), normalize(randLambertianDistVec(c, rng) + N), m.Pcr);
    } else if (roll < reflectionWeight + clearcoatWeight + transmissionWeight) {
        winType = 3;
        outDir = refract(D, N, eta);
    } else {
        if (subsurfaceWeight > 0.0f) {
            if (random_(c, rng) < subsurfaceWeight) { winType = 4; outDir = normalize(randLambertianDistVec(c, rng) + N); }
            else { winType = 0; outDir = normalize(randLambertianDistVec(c, rng) + N); }
        } else { winType = 0; outDir = normalize(randLambertianDistVec(c, rng) + N); }
    }
    return outDir;
}

// frag.glsl:810-882
vec3 trace(Ctx& c, Inv& g, vec3 o, vec3 d, uint32_t& rng) {
    vec3 O = o, D = d, col = v3(1.0f), incLight = v3(0.0f);
    clearIndiceStack(g);
    addToIndiceStack(g, 1.0029f);
    g.RAY_IN_OBJECT = false;
    int bounce = 0;
    while ((float)bounce < c.MAX_BOUNCES) {
        bounce++;
        if (c.count) c.cnt[C_SEGMENTS]++;
        Hit hit = rayScene(c, O, D);
        if (hit.id > -1) {
            O = hit.loc;
            Mtl m = mapMtl(c, newMtl(c, hit.material), hit.uvx, hit.uvy);                       // :825-826
            vec3 N = (m.map_norm > -1) ? sampleTexture(c, m.map_norm, hit.uvx, hit.uvy) : hit.norm;   // :827 raw texel, no tangent frame
            vec3 emission = m.Ke;
            float ND = dot(N, D);
            N = N * (ND > 0.0f ? -1.0f : 1.0f);
            float n1, n2;
            if (ND < 0.0f) {
                addToIndiceStack(g, m.Ni);
                n1 = g.stack[1]; n2 = g.stack[0];
            } else {
                n1 = g.stack[0]; n2 = g.stack[1];
                removeFirstOfIndiceStack(g);
            }
            int w = 0;
            vec3 newD = chooseRay(c, m, n1, n2, N, D, rng, w);
            bool isSpecular = (w == 2);
            D = newD;
            if (w == 3) {
                if (ND < 0.0f) {
                    if (g.RAY_IN_OBJECT) { g.DISTANCE_TRAVELED = distance(g.RAY_ENTER_LOCATION, O); g.APPLY_ABSORBTION = true; }
                    g.RAY_IN_OBJECT = true;
                    g.RAY_ENTER_LOCATION = O;
                } else {
                    g.RAY_IN_OBJECT = false;
                    g.DISTANCE_TRAVELED = distance(g.RAY_ENTER_LOCATION, O);
                    g.APPLY_ABSORBTION = true;
                }
            }
            incLight = incLight + emission * col;
            if (length(col) < 0.1f) return incLight;
            if (g.APPLY_ABSORBTION) {
                col = col * exp3((-m.Tf) * g.DISTANCE_TRAVELED * m.Density);
                g.APPLY_ABSORBTION = false;
            } else if (w == 4) {
            } else {
                col = col * (isSpecular ? m.Ks : m.Kd);
            }
        } else {
            incLight = incLight + bgCol(c, D) * col;
            break;
        }
    }
    return incLight;
}

// frag.glsl:655-681: the RAYTRACING == 0 mode.  One rayScene per sample; fixed up-light shading
// col = Ka + 0.2 Kd + Kd * N.y + Ke (N is NOT flipped towards the ray); with subsurface > 0 the colour is replaced by
// exp(-si / max(radius, 1e-4)) * subsurfaceColor where si = distance(o, rayBVH(hit.loc, d, hit.parentID, 1e30).loc):
// that `.loc` is rayBVH's (t,u,v) triple, not a position (:493), the probe starts ON the surface without the 1e-4 offset
// (rayBVH is called directly, :668), and an ellipsoid hit has parentID = -1 (out-of-bounds BVH read in the shader:
// the restatement treats that probe as a miss, loc = 1e30).
vec3 directDiffuse(Ctx& c, vec3 o, vec3 d) {
    if (c.count) c.cnt[C_SEGMENTS]++;
    Hit hit = rayScene(c, o, d);
    if (hit.id > -1) {
        Mtl m = mapMtl(c, newMtl(c, hit.material), hit.uvx, hit.uvy);                           // :658-659
        vec3 N = (m.map_norm > -1) ? sampleTexture(c, m.map_norm, hit.uvx, hit.uvy) : hit.norm;       // :660
        vec3 col = m.Ka + m.Kd * 0.2f + (m.Kd * dot(v3(0.0f, 1.0f, 0.0f), N)) + m.Ke;
        if (m.subsurface > 0.0f) {
            vec3 loc = v3(1e30f);
            if (hit.parentID >= 0) {
                BvhResult r = rayBVH(c, hit.loc, d, hit.parentID, 1e30f);
                if (r.any) loc = v3(r.t, r.u, r.v);
            }
            float si = distance(o, loc);
            vec3 rad = v3(maxnum(m.subsurfaceRadius.x, 1e-4f), maxnum(m.subsurfaceRadius.y, 1e-4f), maxnum(m.subsurfaceRadius.z, 1e-4f));
            vec3 sigma_t = v3(1.0f / rad.x, 1.0f / rad.y, 1.0f / rad.z);
            col = exp3((-sigma_t) * si) * m.subsurfaceColor;
        }
        return col;
    }
    return bgCol(c, d);
}

// frag.glsl:539-547: the DEBUG traversal heat-map.  rayBVH's .col (:534) per BVH = outputColor*0.1 + (0,0,exp(0.01*(boxTests-200))) +
// (exp(0.02*(triTests-150)),0,0): outputColor gains (0.1,0,0) per leaf node popped (:478), boxTests 2 per inner node popped (:521) and
// triTests is never incremented.  Every BVH is traversed on its own from closest_t = 1e30 (a root miss returns 1e30, and 1e30 > 1e30 is
// false, so even a missed root is popped once), with the un-offset origin and the un-normalised primary direction.
vec3 debugRayScene(Ctx& c, vec3 o, vec3 d) {
    const orc_scene* s = c.s;
    const int numObj = s->obj_indices[0];
    vec3 ret = v3(0.0f);
    const bool wasCounting = c.count;
    uint64_t saved[C_N];
    for (int k = 0; k < C_N; k++) saved[k] = c.cnt[k];
    for (int I = 1; I < numObj + 1; I++) {
        c.count = true; c.cnt[C_BOXTESTS] = 0; c.cnt[C_NODES] = 0;
        rayBVH(c, o, d, s->obj_indices[I], 1e30f);
        const int boxTests = (int)c.cnt[C_BOXTESTS] - 1;               // without the root test of :468
        const int leaves = (int)c.cnt[C_NODES] - boxTests / 2;
        float ocx = 0.0f;
        for (int k = 0; k < leaves; k++) ocx = ocx + 0.1f;             // outputColor += vec3(0.1,0,0)
        const int triTests = 0;
        vec3 col = v3(ocx * 0.1f + 0.0f + exp_(0.02f * (float)(triTests - 150)), 0.0f * 0.1f + 0.0f + 0.0f, 0.0f * 0.1f + exp_(0.01f * (float)(boxTests - 200)) + 0.0f);
        ret = ret + v3(col.x / (float)numObj, col.y / (float)numObj, col.z / (float)numObj);
    }
    c.count = wasCounting;
    for (int k = 0; k < C_N; k++) c.cnt[k] = saved[k];
    return ret;
}

// frag.glsl:884-934 for one pixel.  texCoord is the pixel centre (vert.glsl:14 interpolated).
void shadePixel(Ctx& c, int px, int py, int W, int H, int u_frameCount, int u_seed, float mid_to_scene, float* FRAME) {
    float tcx = ((float)px + 0.5f) / (float)W, tcy = ((float)py + 0.5f) / (float)H;
    float resY = c.resolution * c.screenHratio;
    int pcx = (int)(tcx * c.resolution), pcy = (int)(tcy * resY);
    uint32_t index = (uint32_t)pcy * (uint32_t)c.resolution + (uint32_t)pcx;
    if (pcx >= (int)c.resolution || pcy >= (int)resY) return;
    if (__builtin_fabsf((float)pcx - c.MOUSE_POS.x) < c.resolution * 0.005f &&
        __builtin_fabsf((float)pcy - c.MOUSE_POS.y) < c.resolution * 0.005f)
        return;   // mouse-probe overlay (frag.glsl:888-893) writes fragColor only; FRAME untouched
    vec3 p = v3(((tcx * 2.0f - 1.0f) * -1.0f) * c.screenSize, ((tcy * 2.0f - 1.0f) * c.screenHratio) * c.screenSize, c.focalLength);
    vec3 direction = vecmat(p, c.camRot);
    vec3 col = v3(0.0f);
    uint32_t rng = index + (uint32_t)u_seed;
    Inv g;
    if (c.DEBUG != 0.0f) {                                            // :916-918
        col = debugRayScene(c, c.ORIGIN, direction);
        float* F = FRAME + 4 * ((int64_t)pcy * W + pcx);
        if ((float)u_frameCount == 1.0f) { F[0] = col.x; F[1] = col.y; F[2] = col.z; F[3] = 1.0f; }
        else { F[0] = F[0] + col.x; F[1] = F[1] + col.y; F[2] = F[2] + col.z; F[3] = F[3] + 1.0f; }
        return;
    }
    for (int rayID = 0; (float)rayID < c.SAMPLE_RES; rayID++) {
        vec3 origin_jittered = c.ORIGIN + vecmat(randLambertianDistVec(c, rng) * c.BLUR, c.camRot);
        float internal_focal_distance = c.FOCAL_DISTANCE;
        if (c.AUTO_FOCUS == 1.0f) { if (mid_to_scene > 0.0f) internal_focal_distance = mid_to_scene; }
        vec3 focal_point = c.ORIGIN + direction * internal_focal_distance;
        vec3 direction_adjusted = normalize(focal_point - origin_jittered);
        if (c.count) c.cnt[C_SAMPLES]++;
        if (c.RAYTRACING == 1.0f) col = col + trace(c, g, origin_jittered, direction_adjusted, rng);
        else col = col + directDiffuse(c, origin_jittered, direction_adjusted);          // :909-913
    }
    float inv = c.SAMPLE_RES;
    col = v3(col.x / inv, col.y / inv, col.z / inv);
    float* F = FRAME + 4 * ((int64_t)pcy * W + pcx);
    if ((float)u_frameCount == 1.0f) { F[0] = col.x; F[1] = col.y; F[2] = col.z; F[3] = 1.0f; }
    else { F[0] = F[0] + col.x; F[1] = F[1] + col.y; F[2] = F[2] + col.z; F[3] = F[3] + 1.0f; }
}

int setupCtx(Ctx& c, const orc_scene* s) {
    c.s = s;
    const float* P = s->params;
    c.screenSize = P[0]; c.focalLength = P[1]; c.resolution = P[2]; c.screenHratio = P[3]; c.SAMPLE_RES = P[4];
    c.MAX_BOUNCES = P[5]; c.GAMMA = P[6]; c.BLUR = P[7]; c.FOCAL_DISTANCE = P[8]; c.RAYTRACING = P[9]; c.DEBUG = P[10]; c.AUTO_FOCUS = P[11];
    c.ORIGIN = v3(s->origin[0], s->origin[1], s->origin[2]);
    c.ROTATION = v3(s->rotation[0], s->rotation[1], s->rotation[2]);
    c.MOUSE_POS = v3(s->mouse[0], s->mouse[1], s->mouse[2]);
    c.me = (int)s->mtl[0];
    c.numObj = s->obj_indices[0];
    c.numImplicits = (int)s->imp[0];
    c.numEllipsoids = (int)s->ellip[0];
    c.camRot = rotationMatrix(c.ROTATION);
    c.count = true;
    if (c.numImplicits < 0) return -3;                           // (implicits: looped over and never hit, see intersectScene)
    int nm = c.me > 0 ? (int)((s->n_mtl_floats - 1) / c.me) : 0;
    for (int m = 0; m < nm; m++) {                               // every texture a material names must have been uploaded
        Mtl t = newMtl(c, m);
        const int maps[] = {t.map_Ka, t.map_Kd, t.map_Ks, t.map_Ke, t.map_d, t.map_Tr, t.map_Ns, t.map_Pm, t.map_Pr, t.map_Ps, t.map_Pc, t.map_norm};
        for (int idx : maps) {
            if (idx <= -1) continue;
            if (idx == 0) continue;
            if (!s->tex || idx >= s->n_tex || !s->tex[idx]) return -4;
        }
    }
    return 0;
}

}  // namespace

extern "C" {

// One frame (one glDrawArrays, dispatch.java:705) over pixels x0 + i*xs, y0 + j*ys.
// frame: W*H*4 f32 accumulator (FRAME image), updated in place.  counters: C_N u64, summed.
int orc_render(const orc_scene* s, int W, int H, int u_frameCount, int u_seed, float* frame, int nthreads,
               int x0, int xs, int y0, int ys, uint64_t* counters) {
    Ctx base;
    int rc = setupCtx(base, s);
    if (rc) return rc;
    if (W != (int)base.resolution || H != (int)(base.resolution * base.screenHratio)) return -5;
    uint64_t dummy[C_N] = {0};
    base.cnt = dummy;
    // auto-focus ray (frag.glsl:901-906): uniform inputs -> identical for every pixel and sample
    float mid = -1.0f;
    if (base.AUTO_FOCUS == 1.0f) {
        base.count = false;
        mid = rayScene(base, base.ORIGIN, vecmat(v3(0, 0, 1), base.camRot)).distance;
        base.count = true;
    }
    if (nthreads < 1) nthreads = 1;
    if (xs < 1) xs = 1;
    if (ys < 1) ys = 1;
    std::vector<std::vector<uint64_t>> tc(nthreads, std::vector<uint64_t>(C_N, 0));
    std::atomic<int> nextRow{0};
    int nrows = (H - y0 + ys - 1) / ys;
    auto work = [&](int tid) {
        Ctx c = base;
        c.cnt = tc[tid].data();
        for (;;) {
            int r = nextRow.fetch_add(1);
            if (r >= nrows) break;
            int y = y0 + r * ys;
            for (int x = x0; x < W; x += xs) shadePixel(c, x, y, W, H, u_frameCount, u_seed, mid, frame);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    if (counters)
        for (int t = 0; t < nthreads; t++)
            for (int k = 0; k < C_N; k++) counters[k] += tc[t][k];
    return 0;
}

// distance of the auto-focus ray (-1 on miss): exposed for tests
float orc_autofocus(const orc_scene* s) {
    Ctx c; if (setupCtx(c, s)) return NAN;
    uint64_t dummy[C_N] = {0}; c.cnt = dummy; c.count = false;
    return rayScene(c, c.ORIGIN, vecmat(v3(0, 0, 1), c.camRot)).distance;
}

// single-ray probe for tests: returns hit id (-1 miss), fills out[0..7] = t, loc.xyz, norm.xyz, material
int orc_ray_scene(const orc_scene* s, const float* o, const float* d, float* out) {
    Ctx c; if (setupCtx(c, s)) return -100;
    uint64_t dummy[C_N] = {0}; c.cnt = dummy; c.count = false;
    Hit h = rayScene(c, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
    out[0] = h.distance; out[1] = h.loc.x; out[2] = h.loc.y; out[3] = h.loc.z;
    out[4] = h.norm.x; out[5] = h.norm.y; out[6] = h.norm.z; out[7] = (float)h.material;
    return h.id > -1 ? (h.type * 0x1000000 + h.id) : -1;
}

// RNG known-answer helper: advances *state n times, writes results and random() floats
void orc_rng(uint32_t* state, int n, uint32_t* results, float* randoms) {
    for (int i = 0; i < n; i++) {
        uint32_t r = NextRandom(*state);
        if (results) results[i] = r;
        if (randoms) randoms[i] = (float)r / 4294967295.0f;
    }
}

// math contract probes: fn 0 sin,1 cos,2 log,3 exp,4 atan2(x,y),5 asin
void orc_math(int fn, const float* x, const float* y, float* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = sin_(x[i]); break;
            case 1: out[i] = cos_(x[i]); break;
            case 2: out[i] = log_(x[i]); break;
            case 3: out[i] = exp_(x[i]); break;
            case 4: out[i] = atan2_(x[i], y[i]); break;
            case 5: out[i] = asin_(x[i]); break;
            default: out[i] = NAN;
        }
    }
}

void orc_rotate(const float* p, const float* rot, int back, float* out) {
    vec3 r = back ? rotateBack(v3(p[0], p[1], p[2]), v3(rot[0], rot[1], rot[2])) : rotate(v3(p[0], p[1], p[2]), v3(rot[0], rot[1], rot[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

// 8-bit display path = what the reference's screenshot writes:
//   fragColor = newTotal / frameCount (frag.glsl:932; vec4(col,1) for frame 1, :927)  -> default framebuffer, UNORM8:
//   clamp to [0,1], *255, round to nearest (NaN -> 0)  -> glReadPixels(GL_RGB, GL_UNSIGNED_BYTE) (dispatch.java:813)
//   -> pixel = (r << 16) + (g << 8) + b with SIGNED Java bytes (:819-822): a channel >= 128 borrows 1 from the channel
//      above it (Q-18, reproduced when java_bytes != 0)  -> vertical flip (:828-833) -> rows top first.
// The Java2D bilinear AffineTransformOp used for the flip maps pixel centres onto pixel centres; its edge/alpha
// behaviour is JRE-specific and not restated.
void orc_display(const float* frame, int W, int H, int frameCount, int java_bytes, uint8_t* rgb_out) {
    float fc = (float)frameCount;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const float* F = frame + 4 * ((int64_t)y * W + x);
            int q[3];
            for (int k = 0; k < 3; k++) {
                float v = F[k] / fc;
                v = (v != v) ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
                q[k] = (int)__builtin_floorf(v * 255.0f + 0.5f);
            }
            int r = q[0], g = q[1], b = q[2];
            if (java_bytes) {
                int32_t pix = (int32_t)((uint32_t)(int32_t)(int8_t)r << 16) + (int32_t)((uint32_t)(int32_t)(int8_t)g << 8) + (int32_t)(int8_t)b;
                r = (pix >> 16) & 0xff; g = (pix >> 8) & 0xff; b = pix & 0xff;
            }
            uint8_t* o = rgb_out + 3 * ((int64_t)(H - 1 - y) * W + x);
            o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
        }
    }
}

// what-if statistics of the object loop (0 = off); see xObjectLoop
void orc_set_whatif(int mode, float margin) { g_whatif = mode; g_whatif_margin = margin; }

int orc_has_fma(void) { return __builtin_cpu_supports("fma") ? 1 : 0; }

}  // extern "C"
