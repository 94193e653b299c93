/* include/pt_scene.h — C ABI of the host-side scene producer (libpt_host.so).
 *
 * The reference's host is Java (src/Main/dispatch.java) and no JDK exists in this environment,
 * so the producers of the hot path's input buffers are mirrored in C++ behind this C ABI:
 * same operations, argument meaning and error behaviour as the reference's scene DSL.
 * Every entry point cites the reference method it replaces.  All functions return 0 on success
 * and a negative code on error; the message is available from pts_last_error() (the reference
 * throws RuntimeException / IndexOutOfBoundsException at the same points).
 * Not thread-safe per scene (the reference is single-threaded, dispatch.java:168).
 */
#ifndef PT_SCENE_H
#define PT_SCENE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pts_scene pts_scene;

/* the reference's static scene lists (dispatch.java:94-131) as one object */
pts_scene* pts_create(void);
void pts_destroy(pts_scene* s);
const char* pts_last_error(void);

/* scene.addMaterial(name)                                         dispatch.java:1048-1052
 * (material() defaults, :1514-1550) -> returns the material index (>= 0) */
int pts_add_material(pts_scene* s, const char* name);
/* scene.setLastMtl(property, val)                                 dispatch.java:1054-1062
 * n = 1 for double/int fields, n = 3 for vec fields; unknown property -> error
 * ("Not a valid property"), wrong arity -> error (IllegalArgumentException in Java). */
int pts_set_last_mtl(pts_scene* s, const char* property, const double* val, int n);

/* textures.add(path); textureNames.add(name)                       dispatch.java:221-222 (the sky is entry 0), :1570-1571
 * -> index of the new entry in the texture table (binding 15).  The host layer decodes the files (the reference: stbi_load,
 * dispatch.java:343) and hands the pixels to pt_set_texture(index, ...). */
int pts_add_texture(pts_scene* s, const char* path, const char* name);
int pts_texture_count(pts_scene* s);
const char* pts_texture_path(pts_scene* s, int index);
const char* pts_texture_name(pts_scene* s, int index);

/* material.parseMtls(filePath, parentDirectoryPath)                dispatch.java:1319-1512
 * One material per `newmtl` block (a block ends at the first empty line), named <name><parentDirectoryPath>; keys Ka Kd Ks Ns
 * d (sets Tr = 1-d) Tr (sets d = 1-Tr) Tf Ni Ke (sets Density = |Ke|) Density illum Pm Pr Ps Pc Pcr aniso anisor subsurface
 * subsurfaceColor subsurfaceRadius and the maps map_Ka map_Kd map_Ks map_Pm map_Pr|refl map_Ps map_Pc map_Pcr
 * map_Bump|bump|map_bump map_d map_Tr map_Ns map_Ke, each registering <parentDirectoryPath>/<file> in the texture table unless
 * a texture of that name is already there.  Lines are split on single spaces, as the Java does. */
int pts_parse_mtls(pts_scene* s, const char* mtl_path, const char* parent_directory);

/* Replace the recursive CPU builder (BVH(int,int), dispatch.java:1630-1752) by an external one with the signature of
 * pt_build_bvh (include/pt_api.h) for the objects added from now on; fn == NULL restores the built-in builder.  The library
 * itself stays free of any GPU dependency: the caller passes the function pointer. */
typedef int (*pts_bvh_builder)(int device, const double* tri9, int64_t n_tris, int32_t* n_nodes, double* node_bounds, int32_t* node_links,
                               int32_t* node_leaf, int32_t* leaf_tris, int32_t* max_depth);
int pts_set_bvh_builder(pts_scene* s, pts_bvh_builder fn, int device, const char* (*last_error)(void));

/* scene.addObject(filepath, material, scale, shift, rot) for a regular .obj file
 *                                                                  dispatch.java:867-886, 888-1003
 * parent_directory may be NULL (then "usemtl X" looks for a material named "Xnull", exactly
 * as the Java string concatenation does, :924). One BVH per o/g group (:907-921, :993-997).
 * When obj_path is a DIRECTORY (:869-882): every *.mtl in it goes through pts_parse_mtls(file, dir) and every *.obj through the
 * OBJ parser with parentDirectory = dir (files in alphabetical order; Java's listFiles order is unspecified). */
int pts_add_object(pts_scene* s, const char* obj_path, int material, const double scale[3],
                   const double shift[3], const double rot[3], const char* parent_directory);
/* same parser fed from memory (procedural meshes) */
int pts_add_object_text(pts_scene* s, const char* obj_text, size_t len, int material, const double scale[3],
                        const double shift[3], const double rot[3], const char* parent_directory);

/* scene.addTri(v1,v2,v3,m)                                         dispatch.java:1013-1015
 * (normals become NaN: new vec(0).normalize(); such triangles are in no BVH) */
int pts_add_tri(pts_scene* s, const double v1[3], const double v2[3], const double v3[3], int m);
/* scene.addEllipsoid(c, stretch, rot, radius, m)                   dispatch.java:1017-1023 */
int pts_add_ellipsoid(pts_scene* s, const double c[3], const double stretch[3], const double rot[3], float radius, int m);
/* scene.addImplicit(fn, shift, scale, rot, m)                      dispatch.java:1005-1011 */
int pts_add_implicit(pts_scene* s, int fn, const double shift[3], const double scale[3], const double rot[3], int m);

/* BVH.allBVHtoList() + BVH.sortTree() + the SSBO packers          dispatch.java:270-329, 386-534, 1764-1833
 * Must be called after the last add_* and before pts_get_buffer. */
int pts_pack(pts_scene* s);

/* Packed buffer for an SSBO binding point (3 tris, 5 implicits, 7 ellipsoids, 10 BVHdata,
 * 11 BVHtree, 12 leafTriIndices, 13 objIndices, 14 mtlData).  The pointer stays valid until
 * the next pts_pack / pts_destroy (caller copies, like a direct NIO buffer handed to glBufferData). */
int pts_get_buffer(pts_scene* s, int binding, const void** data, size_t* bytes);

/* counts: 0 triangles, 1 BVH nodes, 2 objects (BVH roots), 3 materials, 4 ellipsoids,
 * 5 leaf index entries, 6 max BVH depth (root = 0), 7 max triangles in one leaf */
int64_t pts_count(pts_scene* s, int what);

#ifdef __cplusplus
}
#endif
#endif
