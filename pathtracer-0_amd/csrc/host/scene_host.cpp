// scene_host.cpp — host-side producers of the hot path's input buffers (libpt_host.so).
//
// C++ mirror of the reference's Java scene DSL, OBJ loader, BVH builder/flattener and SSBO
// packers (/root/reference/src/Main/dispatch.java:866-1062, 1067-1277, 1279-1317, 1514-1550,
// 1579-1833, 270-329, 386-534).  Double precision exactly where Java uses double; narrowing to
// float32 happens at pack time as in Java.  Built with -ffp-contract=off (Java never fuses).
// Value semantics replace the reference's BoundingBox aliasing: same boxes.
#include "../../../include/pt_scene.h"

#include <algorithm>
#include <cmath>
#include <dirent.h>
#include <sys/stat.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

// ---- vec (dispatch.java:1067-1217) ----
struct vec {
    double x = 0, y = 0, z = 0;
    vec() {}
    explicit vec(double v) : x(v), y(v), z(v) {}
    vec(double a, double b, double c) : x(a), y(b), z(c) {}
    vec add(const vec& o) const { return vec(x + o.x, y + o.y, z + o.z); }
    vec sub(const vec& o) const { return vec(x - o.x, y - o.y, z - o.z); }
    vec mult(const vec& o) const { return vec(x * o.x, y * o.y, z * o.z); }
    vec mult(double c) const { return vec(x * c, y * c, z * c); }
    vec div(double c) const { return vec(x / c, y / c, z / c); }
    double get(int axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
    // vec.rotate (:1157-1191): X, then Y, then Z, radians; the Java method also mutates its receiver
    vec rotate(const vec& rot) const {
        double X = x, Y = y, Z = z;
        double cosX = std::cos(rot.x), sinX = std::sin(rot.x);
        double newY = cosX * Y - sinX * Z;
        double newZ = sinX * Y + cosX * Z;
        Y = newY; Z = newZ;
        double cosY = std::cos(rot.y), sinY = std::sin(rot.y);
        double newX = cosY * X + sinY * Z;
        newZ = -sinY * X + cosY * Z;
        X = newX; Z = newZ;
        double cosZ = std::cos(rot.z), sinZ = std::sin(rot.z);
        newX = cosZ * X - sinZ * Y;
        newY = sinZ * X + cosZ * Y;
        return vec(newX, newY, newZ);
    }
    vec cross(const vec& o) const { return vec(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
    double magnitude() const { return std::sqrt(x * x + y * y + z * z); }
    vec normalize() const { double m = magnitude(); return vec(x / m, y / m, z / m); }   // (0,0,0) -> NaN (Q-5)
};

// ---- triangle (dispatch.java:1220-1277) ----
struct triangle {
    vec v1, v2, v3, n1, n2, n3, vt1, vt2, vt3;
    int material = 0;
    int ID = 0;
    vec min, max, centroid;
};
vec vmin3(const vec& a, const vec& b, const vec& c) {
    vec m(std::numeric_limits<double>::max());
    for (const vec* v : {&a, &b, &c}) { if (v->x < m.x) m.x = v->x; if (v->y < m.y) m.y = v->y; if (v->z < m.z) m.z = v->z; }
    return m;
}
vec vmax3(const vec& a, const vec& b, const vec& c) {
    vec m(-std::numeric_limits<double>::infinity());
    for (const vec* v : {&a, &b, &c}) { if (v->x > m.x) m.x = v->x; if (v->y > m.y) m.y = v->y; if (v->z > m.z) m.z = v->z; }
    return m;
}

// ---- material (dispatch.java:1279-1317, defaults :1514-1550) ----
struct material {
    std::string name;
    vec Ka{0}, Kd{0.8}, Ks{0.5};
    double Ns = 10, d = 0, Tr = 0;
    vec Tf{0};
    double Ni = 1;
    vec Ke{0};
    int illum = 0, map_Ka = -1, map_Kd = -1, map_Ks = -1;
    double Pm = 0, Pr = 1, Ps = 0, Pc = 0, Pcr = 0, aniso = 0, anisor = 0;
    int map_Pm = -1, map_Pr = -1, map_Ps = -1, map_Pc = -1, map_Pcr = -1, map_bump = -1, map_d = -1, map_Tr = -1, map_Ns = -1, map_Ke = -1;
    double Density = 1, subsurface = 0;
    vec subsurfaceColor{0}, subsurfaceRadius{0};
};

// ---- BVH (dispatch.java:1579-1842) ----
// java.lang.Math.min/max on doubles: NaN wins, -0.0 < +0.0
inline double jmin(double a, double b) { if (a != a) return a; if (b != b) return b; if (a == 0.0 && b == 0.0) return std::signbit(a) ? a : b; return a < b ? a : b; }
inline double jmax(double a, double b) { if (a != a) return a; if (b != b) return b; if (a == 0.0 && b == 0.0) return std::signbit(a) ? b : a; return a > b ? a : b; }
struct BoundingBox {
    vec Min, Max, Size;
    bool hasPoint = false;
    void Grow(const vec& mn, const vec& mx) {                                   // GrowToInclude :1612-1627
        if (hasPoint) {
            Min.x = jmin(mn.x, Min.x); Min.y = jmin(mn.y, Min.y); Min.z = jmin(mn.z, Min.z);
            Max.x = jmax(mx.x, Max.x); Max.y = jmax(mx.y, Max.y); Max.z = jmax(mx.z, Max.z);
        } else { hasPoint = true; Min = mn; Max = mx; }
        Size = Max.sub(Min);
    }
};
struct BVH {
    vec min, max;
    std::vector<int> storedTri;
    std::unique_ptr<BVH> Left, Right;
    int branchDepth = 0;
    int ID = 0;
};

constexpr int MAX_BVH_BRANCHES = 256;       // dispatch.java:45
constexpr int MAX_TRIS_IN_BVH_LEAF = 1;     // :46
constexpr int OPTIMIZATION_LEVEL = 5;       // :47
constexpr int NUM_MATERIAL_PARAMETERS = 48; // :97

}  // namespace

struct pts_scene {
    std::vector<material> materials;
    std::vector<std::string> textures, textureNames;      // dispatch.java:95-96
    std::vector<triangle> triangles;
    int NEXT_TRI_ID = 0;
    // implicits
    std::vector<int> fn; std::vector<vec> Ishift, Irot, Iscale; std::vector<int> Im;
    // ellipsoids
    std::vector<vec> Ec, Estretch, Erot; std::vector<float> Erad; std::vector<int> Em;
    // BVHs
    std::vector<std::unique_ptr<BVH>> sceneObjs;
    int nextBVHId = 0;
    pts_bvh_builder builder = nullptr; int builderDevice = 0; const char* (*builderError)(void) = nullptr;   // pts_set_bvh_builder
    // packed
    std::vector<float> triBuf, impBuf, ellipBuf, bvhData, mtlBuf;
    std::vector<int32_t> bvhTree, leafTri, objIdx;
    bool packed = false;
    int64_t maxDepth = 0, maxLeaf = 0;
};

namespace {

triangle makeTriangle(pts_scene* s, const vec& v1, const vec& v2, const vec& v3, const vec& n1, const vec& n2, const vec& n3,
                      const vec& vt1, const vec& vt2, const vec& vt3, int material) {   // triangle ctor :1237-1255
    triangle t;
    t.v1 = v1; t.v2 = v2; t.v3 = v3;
    t.n1 = n1.normalize(); t.n2 = n2.normalize(); t.n3 = n3.normalize();
    t.vt1 = vt1; t.vt2 = vt2; t.vt3 = vt3;
    t.material = material;
    t.min = vmin3(v1, v2, v3); t.max = vmax3(v1, v2, v3);
    t.centroid = (v1.add(v2.add(v3))).div(3);
    t.ID = s->NEXT_TRI_ID++;
    return t;
}

double cost(const vec& extent, int numTri) {                                        // :1748-1752
    if (numTri == 0) return std::numeric_limits<double>::infinity();
    double halfSurfaceArea = (extent.x * extent.y + extent.x * extent.z + extent.y * extent.z);
    return std::fabs(halfSurfaceArea) * (double)numTri;
}
double testSplit(int axis, double pos, const std::vector<const triangle*>& tris) {  // testSplitOnTEST :1722-1747
    int numLeft = 0, numRight = 0;
    BoundingBox lb, rb;
    for (const triangle* t : tris) {
        if (t->centroid.get(axis) < pos) { lb.Grow(t->min, t->max); numLeft++; }
        else { rb.Grow(t->min, t->max); numRight++; }
    }
    double L = numLeft == 0 ? std::numeric_limits<double>::infinity() : cost(lb.Size, numLeft);
    double R = numRight == 0 ? std::numeric_limits<double>::infinity() : cost(rb.Size, numRight);
    return L + R;
}

// splitTEST :1647-1721.  Returns false when no split is taken (Java: empty children list).
bool splitNode(pts_scene* s, const BoundingBox& bounds, const std::vector<const triangle*>& tris, double bestCost, int branchDepth,
               std::unique_ptr<BVH>& outL, std::unique_ptr<BVH>& outR) {
    int bestAxis = 0;
    double bestPos = -1;
    for (int axis = 0; axis < 3; axis++) {
        for (int i = 0; i < OPTIMIZATION_LEVEL; i++) {
            double splitPercent = ((double)i + 1.0) / (OPTIMIZATION_LEVEL + 1.0);
            double pos = bounds.Min.get(axis) + bounds.Size.get(axis) * splitPercent;
            double c = testSplit(axis, pos, tris);
            if (c < bestCost) { bestCost = c; bestAxis = axis; bestPos = pos; }
        }
    }
    if (bestPos == -1) return false;                      // sentinel doubles as "no split" (Q-11)
    std::vector<const triangle*> leftTris, rightTris;
    std::vector<int> leftIDs, rightIDs;
    BoundingBox lb, rb;
    for (const triangle* t : tris) {
        if (t->centroid.get(bestAxis) < bestPos) { leftTris.push_back(t); lb.Grow(t->min, t->max); leftIDs.push_back(t->ID); }
        else { rightTris.push_back(t); rb.Grow(t->min, t->max); rightIDs.push_back(t->ID); }
    }
    auto build = [&](std::vector<const triangle*>& side, std::vector<int>& ids, BoundingBox& bb, std::unique_ptr<BVH>& out) {
        if (side.empty() || side.size() == tris.size()) { out.reset(); return; }      // children.add(null)
        out.reset(new BVH());
        out->ID = s->nextBVHId++;                                                       // BVH(vec,vec,List,int) :1755-1762
        out->min = bb.Min; out->max = bb.Max; out->branchDepth = branchDepth + 1;
        if (branchDepth >= MAX_BVH_BRANCHES || (int)side.size() <= MAX_TRIS_IN_BVH_LEAF) {
            out->storedTri = ids;
        } else {
            std::unique_ptr<BVH> l, r;
            if (splitNode(s, bb, side, bestCost, branchDepth + 1, l, r)) { out->Left = std::move(l); out->Right = std::move(r); }
            if (!out->Left && !out->Right) out->storedTri = ids;
        }
    };
    build(leftTris, leftIDs, lb, outL);
    build(rightTris, rightIDs, rb, outR);
    return true;
}

// the same object through an external builder (pt_build_bvh): rebuild the node objects from its pre-order arrays
int buildObjectBVHExternal(pts_scene* s, int start, int end) {
    const int64_t n = end - start;
    std::vector<double> tri9((size_t)n * 9);
    for (int64_t i = 0; i < n; i++) {
        const triangle& t = s->triangles[(size_t)start + i];
        double* o = &tri9[(size_t)i * 9];
        o[0] = t.min.x; o[1] = t.min.y; o[2] = t.min.z; o[3] = t.max.x; o[4] = t.max.y; o[5] = t.max.z; o[6] = t.centroid.x; o[7] = t.centroid.y; o[8] = t.centroid.z;
    }
    std::vector<double> bounds((size_t)n * 12); std::vector<int32_t> links((size_t)n * 4), leaf((size_t)n * 4), order((size_t)n);
    int32_t nNodes = 0, depth = 0;
    int rc = s->builder(s->builderDevice, tri9.data(), n, &nNodes, bounds.data(), links.data(), leaf.data(), order.data(), &depth);
    if (rc) return fail(-20, std::string("external BVH builder: ") + (s->builderError ? s->builderError() : "failed"));
    std::vector<std::unique_ptr<BVH>> nodes((size_t)nNodes);
    const int idBase = s->nextBVHId;
    for (int k = nNodes - 1; k >= 0; k--) {                      // children have larger pre-order ids than their parent
        std::unique_ptr<BVH> b(new BVH());
        b->ID = idBase + k;
        b->min = vec(bounds[6 * (size_t)k], bounds[6 * (size_t)k + 1], bounds[6 * (size_t)k + 2]);
        b->max = vec(bounds[6 * (size_t)k + 3], bounds[6 * (size_t)k + 4], bounds[6 * (size_t)k + 5]);
        int l = links[2 * (size_t)k], r = links[2 * (size_t)k + 1];
        if (l >= 0) {
            if (l <= k || r <= k || l >= nNodes || r >= nNodes || !nodes[(size_t)l] || !nodes[(size_t)r]) return fail(-25, "external BVH builder returned an inconsistent tree");
            b->Left = std::move(nodes[(size_t)l]); b->Right = std::move(nodes[(size_t)r]);
        } else {
            int a = leaf[2 * (size_t)k], e = leaf[2 * (size_t)k + 1];
            if (a < 0 || e > n || a > e) return fail(-25, "external BVH builder returned an inconsistent leaf range");
            for (int q = a; q < e; q++) b->storedTri.push_back(s->triangles[(size_t)start + order[(size_t)q]].ID);
        }
        nodes[(size_t)k] = std::move(b);
    }
    s->nextBVHId += nNodes;
    s->sceneObjs.push_back(std::move(nodes[0]));
    return 0;
}

// BVH(int triIndicesStart, int triIndicesEnd) :1630-1646
int buildObjectBVH(pts_scene* s, int start, int end) {
    if (s->builder) return buildObjectBVHExternal(s, start, end);
    std::unique_ptr<BVH> root(new BVH());
    root->ID = s->nextBVHId++;
    BoundingBox bounds;
    std::vector<const triangle*> tris;
    for (int i = start; i < end; i++) { tris.push_back(&s->triangles[i]); bounds.Grow(s->triangles[i].min, s->triangles[i].max); }
    root->min = bounds.Min; root->max = bounds.Max;
    std::unique_ptr<BVH> l, r;
    if (!splitNode(s, bounds, tris, std::numeric_limits<double>::infinity(), 0, l, r))
        return fail(-20, "BVH root could not be split by any candidate plane (reference: IndexOutOfBoundsException at dispatch.java:1644, SURVEY Q-16)");
    root->Left = std::move(l); root->Right = std::move(r);
    s->sceneObjs.push_back(std::move(root));
    return 0;
}

std::vector<std::string> splitWs(const std::string& line) {       // String.split("\\s+")
    std::vector<std::string> out; std::string cur;
    bool first = true;
    for (size_t i = 0; i <= line.size(); i++) {
        bool ws = i == line.size() || line[i] == ' ' || line[i] == '\t' || line[i] == '\r' || line[i] == '\f' || line[i] == '\v';
        if (ws) {
            if (!cur.empty() || (first && i < line.size())) out.push_back(cur);   // Java keeps a leading empty token
            cur.clear(); first = false;
            while (i + 1 < line.size() && (line[i + 1] == ' ' || line[i + 1] == '\t' || line[i + 1] == '\r')) i++;
        } else { cur.push_back(line[i]); first = false; }
    }
    return out;
}
bool startsWith(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (unsigned char)s[a] <= ' ') a++;
    while (b > a && (unsigned char)s[b - 1] <= ' ') b--;
    return s.substr(a, b - a);
}
bool parseD(const std::string& t, double& out) {
    char* e = nullptr; out = std::strtod(t.c_str(), &e);
    return e && e != t.c_str() && *e == 0;
}
bool parseI(const std::string& t, int& out) {
    char* e = nullptr; long v = std::strtol(t.c_str(), &e, 10); out = (int)v;
    return e && e != t.c_str() && *e == 0;
}

// parseObj :888-1003
int parseObj(pts_scene* s, std::istream& in, int materialArg, const vec& scale, const vec& shift, const vec& rot, const char* parentDirectory) {
    int nCurrentObjVerts = 0;
    int objectStartTri = (int)s->triangles.size();
    int mtl = -1;
    std::vector<vec> vertices{vec(0)}, normals{vec(0)}, texcoords{vec(69.420, 0, 0)};
    std::string line;
    int lineNo = 0;
    while (std::getline(in, line)) {
        lineNo++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (startsWith(line, "o ") || startsWith(line, "g ")) {
            mtl = materialArg;
            if ((int)s->triangles.size() > objectStartTri && nCurrentObjVerts > 0) {
                int rc = buildObjectBVH(s, objectStartTri, (int)s->triangles.size());
                if (rc) return rc;
                nCurrentObjVerts = 0;
            }
            objectStartTri = (int)s->triangles.size();
            continue;
        }
        if (startsWith(line, "usemtl ")) {
            std::string rest = line.substr(line.find(' ') + 1);
            size_t sp = rest.find(' ');
            std::string name = trim(sp == std::string::npos ? rest : rest.substr(0, sp)) + (parentDirectory ? parentDirectory : "null");
            for (size_t i = 0; i < s->materials.size(); i++) if (s->materials[i].name == name) mtl = (int)i;
        } else if (startsWith(line, "v ")) {
            auto p = splitWs(line); double x, y, z;
            if (p.size() < 4 || !parseD(p[1], x) || !parseD(p[2], y) || !parseD(p[3], z)) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": bad vertex");
            vec vf = vec(x, y, z).mult(scale).rotate(rot).add(shift);
            nCurrentObjVerts++;
            vertices.push_back(vf);
        } else if (startsWith(line, "vt ")) {
            auto p = splitWs(line); double x, y;
            if (p.size() < 3 || !parseD(p[1], x) || !parseD(p[2], y)) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": bad vt");
            texcoords.push_back(vec(x, y, 0));
        } else if (startsWith(line, "vn ")) {
            auto p = splitWs(line); double x, y, z;
            if (p.size() < 4 || !parseD(p[1], x) || !parseD(p[2], y) || !parseD(p[3], z)) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": bad vn");
            normals.push_back(vec(x, y, z).mult(scale).rotate(rot));          // Q-10: not inverse-transpose
        } else if (startsWith(line, "f ")) {
            auto parts = splitWs(trim(line).substr(2));
            if (parts.size() < 3) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": face with < 3 vertices");
            int vi[3] = {-1, -1, -1}, ti[3] = {0, 0, 0}, ni[3] = {0, 0, 0};
            for (int i = 0; i < 3; i++) {                                       // first three vertices only (:961)
                std::vector<std::string> comp; std::string cur;
                for (char ch : parts[i]) { if (ch == '/') { comp.push_back(cur); cur.clear(); } else cur.push_back(ch); }
                comp.push_back(cur);
                while (comp.size() > 1 && comp.back().empty()) comp.pop_back();  // Java split drops trailing empties
                if (!comp[0].empty() && !parseI(comp[0], vi[i])) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": bad index");
                if (comp.size() > 1 && !comp[1].empty() && !parseI(comp[1], ti[i])) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": bad index");
                if (comp.size() > 2 && !comp[2].empty() && !parseI(comp[2], ni[i])) return fail(-21, "OBJ line " + std::to_string(lineNo) + ": bad index");
            }
            for (int i = 0; i < 3; i++) {
                if (vi[i] < 0 || vi[i] >= (int)vertices.size() || ti[i] < 0 || ti[i] >= (int)texcoords.size() || ni[i] < 0 || ni[i] >= (int)normals.size())
                    return fail(-22, "OBJ line " + std::to_string(lineNo) + ": index out of range (reference: IndexOutOfBoundsException at dispatch.java:983)");
            }
            s->triangles.push_back(makeTriangle(s, vertices[vi[0]], vertices[vi[1]], vertices[vi[2]], normals[ni[0]], normals[ni[1]], normals[ni[2]],
                                                texcoords[ti[0]], texcoords[ti[1]], texcoords[ti[2]], mtl));
        }
    }
    if ((int)s->triangles.size() > objectStartTri && nCurrentObjVerts > 0) {
        int rc = buildObjectBVH(s, objectStartTri, (int)s->triangles.size());
        if (rc) return rc;
    }
    s->packed = false;
    return 0;
}

// flattenBVH :1786-1816
void flatten(pts_scene* s, const BVH* node, int depth) {
    if (!node) return;
    if (depth > s->maxDepth) s->maxDepth = depth;
    if (!node->Left && !node->Right) {
        int startIdx = (int)s->leafTri.size();
        s->leafTri.insert(s->leafTri.end(), node->storedTri.begin(), node->storedTri.end());
        s->bvhData[8 * node->ID + 6] = (float)startIdx;
        s->bvhData[8 * node->ID + 7] = (float)s->leafTri.size();
        if ((int64_t)node->storedTri.size() > s->maxLeaf) s->maxLeaf = (int64_t)node->storedTri.size();
    }
    s->bvhData[8 * node->ID + 0] = (float)node->min.x; s->bvhData[8 * node->ID + 1] = (float)node->min.y; s->bvhData[8 * node->ID + 2] = (float)node->min.z;
    s->bvhData[8 * node->ID + 3] = (float)node->max.x; s->bvhData[8 * node->ID + 4] = (float)node->max.y; s->bvhData[8 * node->ID + 5] = (float)node->max.z;
    s->bvhTree.push_back(node->ID);
    s->bvhTree.push_back(node->Left ? node->Left->ID : -1);
    s->bvhTree.push_back(node->Right ? node->Right->ID : -1);
    flatten(s, node->Left.get(), depth + 1);
    flatten(s, node->Right.get(), depth + 1);
}

void put3(std::vector<float>& b, const vec& v) { b.push_back((float)v.x); b.push_back((float)v.y); b.push_back((float)v.z); }
void put4(std::vector<float>& b, double x, double y, double z) { b.push_back((float)x); b.push_back((float)y); b.push_back((float)z); b.push_back(0.0f); }

struct Field { const char* name; int kind; size_t off; };   // kind 0 double, 1 int, 2 vec
#define FD(n) {#n, 0, offsetof(material, n)}
#define FI(n) {#n, 1, offsetof(material, n)}
#define FV(n) {#n, 2, offsetof(material, n)}
const Field kFields[] = {FV(Ka), FV(Kd), FV(Ks), FD(Ns), FD(d), FD(Tr), FV(Tf), FD(Ni), FV(Ke), FI(illum), FI(map_Ka), FI(map_Kd), FI(map_Ks),
                         FD(Pm), FD(Pr), FD(Ps), FD(Pc), FD(Pcr), FD(aniso), FD(anisor), FI(map_Pm), FI(map_Pr), FI(map_Ps), FI(map_Pc), FI(map_Pcr),
                         FI(map_bump), FI(map_d), FI(map_Tr), FI(map_Ns), FI(map_Ke), FD(Density), FD(subsurface), FV(subsurfaceColor), FV(subsurfaceRadius)};

std::vector<std::string> splitSpace(const std::string& line) {         // String.split(" "): single spaces, trailing empties dropped
    std::vector<std::string> out; std::string cur;
    for (char ch : line) { if (ch == ' ') { out.push_back(cur); cur.clear(); } else cur.push_back(ch); }
    out.push_back(cur);
    while (!out.empty() && out.back().empty()) out.pop_back();
    return out;
}

// material.parseMtls :1319-1512
int parseMtls(pts_scene* s, const std::string& filePath, const std::string& parentDirectoryPath) {
    std::ifstream in(filePath);
    if (!in) return fail(-40, "cannot open MTL file: " + filePath + " (reference: RuntimeException(IOException), dispatch.java:1510)");
    std::string line;
    auto num = [&](const std::vector<std::string>& v, size_t i, double& out) {
        return i < v.size() && parseD(trim(v[i]), out);
    };
    auto nextLine = [&](std::string& l) { if (!std::getline(in, l)) return false; if (!l.empty() && l.back() == '\r') l.pop_back(); return true; };
    while (nextLine(line)) {
        if (!startsWith(line, "newmtl ")) continue;
        auto parts = splitSpace(line);
        if (parts.size() < 2) return fail(-41, "newmtl without a name");
        material mat;
        mat.name = trim(parts[1]) + parentDirectoryPath;
        while (nextLine(line) && !line.empty()) {
            std::replace(line.begin(), line.end(), '/', '\\');                 // :1330
            auto vals = splitSpace(line);
            auto vec3v = [&](vec& out) { double a, b, c; if (!num(vals, 1, a) || !num(vals, 2, b) || !num(vals, 3, c)) return false; out = vec(a, b, c); return true; };
            auto scal = [&](double& out) { return num(vals, 1, out); };
            auto tex = [&](int& slot) {
                if (vals.size() < 2) return fail(-42, "MTL map line without a file name");
                std::string name = trim(vals[1]);
                auto it = std::find(s->textureNames.begin(), s->textureNames.end(), name);
                if (it != s->textureNames.end()) { slot = (int)(it - s->textureNames.begin()); return 0; }
                slot = (int)s->textures.size();
                std::string file = name; std::replace(file.begin(), file.end(), '\\', '/');      // platform path separator
                std::string path = parentDirectoryPath + "/" + file;
                std::ifstream probe(path);
                if (!probe) return fail(-43, "cannot read texture file: " + path + " (reference: RuntimeException from parseTexture, dispatch.java:1573)");
                s->textureNames.push_back(name); s->textures.push_back(path);                    // parseTexture :1552-1575
                return 0;
            };
            bool ok = true; int rc = 0;
            if (startsWith(line, "Ka ")) ok = vec3v(mat.Ka);
            else if (startsWith(line, "Kd ")) ok = vec3v(mat.Kd);
            else if (startsWith(line, "Ks ")) ok = vec3v(mat.Ks);
            else if (startsWith(line, "Ns ")) ok = scal(mat.Ns);
            else if (startsWith(line, "d ")) { ok = scal(mat.d); mat.Tr = 1 - mat.d; }
            else if (startsWith(line, "Tr ")) { ok = scal(mat.Tr); mat.d = 1 - mat.Tr; }
            else if (startsWith(line, "Tf ")) ok = vec3v(mat.Tf);
            else if (startsWith(line, "Ni ")) ok = scal(mat.Ni);
            else if (startsWith(line, "Ke ")) { ok = vec3v(mat.Ke); mat.Density = mat.Ke.magnitude(); }
            else if (startsWith(line, "Density ")) ok = scal(mat.Density);
            else if (startsWith(line, "illum ")) { int v = 0; ok = vals.size() > 1 && parseI(trim(vals[1]), v); mat.illum = v; }
            else if (startsWith(line, "map_Ka ")) rc = tex(mat.map_Ka);
            else if (startsWith(line, "map_Kd ")) rc = tex(mat.map_Kd);
            else if (startsWith(line, "map_Ks ")) rc = tex(mat.map_Ks);
            else if (startsWith(line, "Pm ")) ok = scal(mat.Pm);
            else if (startsWith(line, "Pr ")) ok = scal(mat.Pr);
            else if (startsWith(line, "Ps ")) ok = scal(mat.Ps);
            else if (startsWith(line, "Pc ")) ok = scal(mat.Pc);
            else if (startsWith(line, "Pcr ")) ok = scal(mat.Pcr);
            else if (startsWith(line, "aniso ")) ok = scal(mat.aniso);
            else if (startsWith(line, "anisor ")) ok = scal(mat.anisor);
            else if (startsWith(line, "map_Pm ")) rc = tex(mat.map_Pm);
            else if (startsWith(line, "map_Pr ") || startsWith(line, "refl")) rc = tex(mat.map_Pr);
            else if (startsWith(line, "map_Ps ")) rc = tex(mat.map_Ps);
            else if (startsWith(line, "map_Pc ")) rc = tex(mat.map_Pc);
            else if (startsWith(line, "map_Pcr ")) rc = tex(mat.map_Pcr);
            else if (startsWith(line, "map_Bump ") || startsWith(line, "bump ") || startsWith(line, "map_bump ")) rc = tex(mat.map_bump);
            else if (startsWith(line, "map_d ")) rc = tex(mat.map_d);
            else if (startsWith(line, "map_Tr ")) rc = tex(mat.map_Tr);
            else if (startsWith(line, "map_Ns ")) rc = tex(mat.map_Ns);
            else if (startsWith(line, "map_Ke ")) rc = tex(mat.map_Ke);
            else if (startsWith(line, "subsurface ")) ok = scal(mat.subsurface);
            else if (startsWith(line, "subsurfaceColor ")) ok = vec3v(mat.subsurfaceColor);
            else if (startsWith(line, "subsurfaceRadius ")) ok = vec3v(mat.subsurfaceRadius);
            if (rc) return rc;
            if (!ok) return fail(-44, "MTL: cannot parse numbers in line '" + line + "' (reference: NumberFormatException)");
        }
        s->materials.push_back(mat);
    }
    s->packed = false;
    return 0;
}

bool endsWithCI(const std::string& s, const char* ext) {
    size_t n = std::strlen(ext);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) if (std::tolower((unsigned char)s[s.size() - n + i]) != ext[i]) return false;
    return true;
}

}  // namespace

extern "C" {

pts_scene* pts_create(void) { return new pts_scene(); }
void pts_destroy(pts_scene* s) { delete s; }
const char* pts_last_error(void) { return g_err.c_str(); }

int pts_add_material(pts_scene* s, const char* name) {
    material m; m.name = name ? name : "";
    s->materials.push_back(m); s->packed = false;
    return (int)s->materials.size() - 1;
}

int pts_set_last_mtl(pts_scene* s, const char* property, const double* val, int n) {
    if (s->materials.empty()) return fail(-10, "setLastMtl: no material (reference: IndexOutOfBoundsException)");
    material& m = s->materials.back();
    for (const Field& f : kFields) {
        if (std::strcmp(f.name, property) != 0) continue;
        char* base = reinterpret_cast<char*>(&m) + f.off;
        if (f.kind == 2) { if (n != 3) return fail(-11, "setLastMtl: vec property needs 3 values (reference: IllegalArgumentException)");
                           *reinterpret_cast<vec*>(base) = vec(val[0], val[1], val[2]); }
        else { if (n != 1) return fail(-11, "setLastMtl: scalar property needs 1 value (reference: IllegalArgumentException)");
               if (f.kind == 0) *reinterpret_cast<double*>(base) = val[0]; else *reinterpret_cast<int*>(base) = (int)val[0]; }
        s->packed = false;
        return 0;
    }
    return fail(-12, "Not a valid property");   // dispatch.java:1060
}

int pts_add_object_text(pts_scene* s, const char* obj_text, size_t len, int material, const double scale[3], const double shift[3],
                        const double rot[3], const char* parent_directory) {
    std::istringstream in(std::string(obj_text, len));
    return parseObj(s, in, material, vec(scale[0], scale[1], scale[2]), vec(shift[0], shift[1], shift[2]), vec(rot[0], rot[1], rot[2]), parent_directory);
}
int pts_set_bvh_builder(pts_scene* s, pts_bvh_builder fn, int device, const char* (*last_error)(void)) {
    s->builder = fn; s->builderDevice = device; s->builderError = last_error;
    return 0;
}
int pts_add_texture(pts_scene* s, const char* path, const char* name) {
    s->textures.push_back(path ? path : ""); s->textureNames.push_back(name ? name : "");
    return (int)s->textures.size() - 1;
}
int pts_texture_count(pts_scene* s) { return (int)s->textures.size(); }
const char* pts_texture_path(pts_scene* s, int i) { return (i >= 0 && (size_t)i < s->textures.size()) ? s->textures[i].c_str() : ""; }
const char* pts_texture_name(pts_scene* s, int i) { return (i >= 0 && (size_t)i < s->textureNames.size()) ? s->textureNames[i].c_str() : ""; }
int pts_parse_mtls(pts_scene* s, const char* mtl_path, const char* parent_directory) { return parseMtls(s, mtl_path, parent_directory ? parent_directory : "null"); }

int pts_add_object(pts_scene* s, const char* obj_path, int material, const double scale[3], const double shift[3], const double rot[3],
                   const char* parent_directory) {
    struct stat sb;
    if (stat(obj_path, &sb) == 0 && S_ISDIR(sb.st_mode)) {                    // scene.addObject on a directory, :869-882
        std::vector<std::string> mtls, objs;
        if (DIR* d = opendir(obj_path)) {
            while (dirent* e = readdir(d)) {
                std::string n = e->d_name;
                if (endsWithCI(n, ".mtl")) mtls.push_back(n); else if (endsWithCI(n, ".obj")) objs.push_back(n);
            }
            closedir(d);
        }
        std::sort(mtls.begin(), mtls.end()); std::sort(objs.begin(), objs.end());
        if (objs.empty()) return fail(-24, "no obj files found in the directory.");
        std::string dir = obj_path;
        for (const std::string& m : mtls) { int rc = parseMtls(s, dir + "/" + m, dir); if (rc) return rc; }
        for (const std::string& o : objs) {
            std::ifstream in(dir + "/" + o);
            if (!in) return fail(-23, "cannot open OBJ file: " + dir + "/" + o);
            int rc = parseObj(s, in, material, vec(scale[0], scale[1], scale[2]), vec(shift[0], shift[1], shift[2]), vec(rot[0], rot[1], rot[2]), dir.c_str());
            if (rc) return rc;
        }
        return 0;
    }
    std::ifstream in(obj_path);
    if (!in) return fail(-23, std::string("cannot open OBJ file: ") + obj_path);   // reference prints the IOException and continues
    return parseObj(s, in, material, vec(scale[0], scale[1], scale[2]), vec(shift[0], shift[1], shift[2]), vec(rot[0], rot[1], rot[2]), parent_directory);
}

int pts_add_tri(pts_scene* s, const double v1[3], const double v2[3], const double v3[3], int m) {
    s->triangles.push_back(makeTriangle(s, vec(v1[0], v1[1], v1[2]), vec(v2[0], v2[1], v2[2]), vec(v3[0], v3[1], v3[2]), vec(0), vec(0), vec(0), vec(0), vec(0), vec(0), m));
    s->packed = false;
    return 0;
}
int pts_add_ellipsoid(pts_scene* s, const double c[3], const double stretch[3], const double rot[3], float radius, int m) {
    s->Ec.push_back(vec(c[0], c[1], c[2])); s->Estretch.push_back(vec(stretch[0], stretch[1], stretch[2])); s->Erot.push_back(vec(rot[0], rot[1], rot[2]));
    s->Erad.push_back(radius); s->Em.push_back(m); s->packed = false;
    return 0;
}
int pts_add_implicit(pts_scene* s, int fn, const double shift[3], const double scale[3], const double rot[3], int m) {
    s->fn.push_back(fn); s->Ishift.push_back(vec(shift[0], shift[1], shift[2])); s->Iscale.push_back(vec(scale[0], scale[1], scale[2]));
    s->Irot.push_back(vec(rot[0], rot[1], rot[2])); s->Im.push_back(m); s->packed = false;
    return 0;
}

int pts_pack(pts_scene* s) {
    // materials :270-329
    s->mtlBuf.clear(); s->mtlBuf.push_back((float)NUM_MATERIAL_PARAMETERS);
    for (const material& m : s->materials) {
        std::vector<float>& b = s->mtlBuf;
        put3(b, m.Ka); put3(b, m.Kd); put3(b, m.Ks);
        b.push_back((float)m.Ns); b.push_back((float)m.d); b.push_back((float)m.Tr); put3(b, m.Tf); b.push_back((float)m.Ni); put3(b, m.Ke);
        b.push_back((float)m.Density); b.push_back((float)m.illum); b.push_back((float)m.map_Ka); b.push_back((float)m.map_Kd); b.push_back((float)m.map_Ks);
        b.push_back((float)m.Pm); b.push_back((float)m.Pr); b.push_back((float)m.Ps); b.push_back((float)m.Pc); b.push_back((float)m.Pcr);
        b.push_back((float)m.aniso); b.push_back((float)m.anisor);
        b.push_back((float)m.map_Pm); b.push_back((float)m.map_Pr); b.push_back((float)m.map_Ps); b.push_back((float)m.map_Pc); b.push_back((float)m.map_Pcr);
        b.push_back((float)m.map_bump); b.push_back((float)m.map_d); b.push_back((float)m.map_Tr); b.push_back((float)m.map_Ns); b.push_back((float)m.map_Ke);
        b.push_back((float)m.subsurface); put3(b, m.subsurfaceColor); put3(b, m.subsurfaceRadius);
    }
    // triangles :386-424
    s->triBuf.clear(); s->triBuf.reserve(40 * s->triangles.size());
    for (const triangle& t : s->triangles) {
        std::vector<float>& b = s->triBuf;
        put4(b, t.v1.x, t.v1.y, t.v1.z); put4(b, t.v2.x, t.v2.y, t.v2.z); put4(b, t.v3.x, t.v3.y, t.v3.z);
        if (!(t.n1.x == 0 && t.n1.y == 0 && t.n1.z == 0)) {
            put4(b, t.n1.x, t.n1.y, t.n1.z); put4(b, t.n2.x, t.n2.y, t.n2.z); put4(b, t.n3.x, t.n3.y, t.n3.z);
        } else {
            vec norm = (t.v3.sub(t.v1)).cross(t.v2.sub(t.v1));
            put4(b, norm.x, norm.y, norm.z); put4(b, 0, 0, 0); put4(b, 0, 0, 0);
        }
        if (t.vt1.x != 69.420 && t.vt1.y != 0) {
            put4(b, t.vt1.x, t.vt1.y, 0); put4(b, t.vt2.x, t.vt2.y, 0); put4(b, t.vt3.x, t.vt3.y, 0);
        } else {
            b.push_back(69.420f); b.push_back(0); b.push_back(0); b.push_back(0); put4(b, 0, 0, 0); put4(b, 0, 0, 0);
        }
        put4(b, (double)t.material, 0, 0);
    }
    // implicits :429-456
    s->impBuf.clear(); s->impBuf.push_back((float)s->fn.size());
    for (int f : s->fn) s->impBuf.push_back((float)f);
    for (const vec& v : s->Ishift) put3(s->impBuf, v);
    for (const vec& v : s->Iscale) put3(s->impBuf, v);
    for (const vec& v : s->Irot) put3(s->impBuf, v);
    for (int m : s->Im) s->impBuf.push_back((float)m);
    // ellipsoids :460-487
    s->ellipBuf.clear(); s->ellipBuf.push_back((float)s->Ec.size());
    for (const vec& v : s->Ec) put3(s->ellipBuf, v);
    for (const vec& v : s->Estretch) put3(s->ellipBuf, v);
    for (const vec& v : s->Erot) put3(s->ellipBuf, v);
    for (float r : s->Erad) s->ellipBuf.push_back(r);
    for (int m : s->Em) s->ellipBuf.push_back((float)m);
    // BVH.allBVHtoList :1764-1785 (+ sortTree :1817-1833: ids are appended in increasing order already; sort kept for fidelity)
    s->leafTri.clear(); s->bvhTree.clear(); s->objIdx.clear();
    s->bvhData.assign((size_t)s->nextBVHId * 8, 0.0f);
    s->maxDepth = 0; s->maxLeaf = 0;
    std::vector<int32_t> roots;
    for (auto& o : s->sceneObjs) { roots.push_back(o->ID); flatten(s, o.get(), 0); }
    {   // sortTree: stable sort of (id,left,right) triples by id
        size_t n = s->bvhTree.size() / 3;
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return s->bvhTree[3 * a] < s->bvhTree[3 * b]; });
        std::vector<int32_t> sorted; sorted.reserve(3 * n);
        for (size_t i : order) { sorted.push_back(s->bvhTree[3 * i]); sorted.push_back(s->bvhTree[3 * i + 1]); sorted.push_back(s->bvhTree[3 * i + 2]); }
        s->bvhTree.swap(sorted);
    }
    s->objIdx.push_back((int32_t)roots.size());                                       // :525-529
    for (int32_t r : roots) s->objIdx.push_back(r);
    s->packed = true;
    return 0;
}

int pts_get_buffer(pts_scene* s, int binding, const void** data, size_t* bytes) {
    if (!s->packed) return fail(-30, "pts_get_buffer before pts_pack");
    switch (binding) {
        case 3: *data = s->triBuf.data(); *bytes = s->triBuf.size() * 4; return 0;
        case 5: *data = s->impBuf.data(); *bytes = s->impBuf.size() * 4; return 0;
        case 7: *data = s->ellipBuf.data(); *bytes = s->ellipBuf.size() * 4; return 0;
        case 10: *data = s->bvhData.data(); *bytes = s->bvhData.size() * 4; return 0;
        case 11: *data = s->bvhTree.data(); *bytes = s->bvhTree.size() * 4; return 0;
        case 12: *data = s->leafTri.data(); *bytes = s->leafTri.size() * 4; return 0;
        case 13: *data = s->objIdx.data(); *bytes = s->objIdx.size() * 4; return 0;
        case 14: *data = s->mtlBuf.data(); *bytes = s->mtlBuf.size() * 4; return 0;
    }
    return fail(-31, "pts_get_buffer: binding is not produced by the scene packers");
}

int64_t pts_count(pts_scene* s, int what) {
    switch (what) {
        case 0: return (int64_t)s->triangles.size();
        case 1: return s->nextBVHId;
        case 2: return (int64_t)s->sceneObjs.size();
        case 3: return (int64_t)s->materials.size();
        case 4: return (int64_t)s->Ec.size();
        case 5: return (int64_t)s->leafTri.size();
        case 6: return s->maxDepth;
        case 7: return s->maxLeaf;
    }
    return -1;
}

}  // extern "C"
