/*
 * rt_amd.h -- C ABI of the MI355X (gfx950) path-tracing backend for the `ray_tracer` crate.
 *
 * This header is the drop-in boundary.  The reference (frederik-kofoed-marqversen/ray_tracing) has
 * no FFI of its own: its operator boundary is the Rust trait `Surface {hit, hit_bool}`
 * (ray_tracer/src/geometry/traits.rs:5-17) and the frame-level pair `Engine::new` / `Engine::render`
 * (ray_tracer/src/render/path_integrator.rs:50, :278).  Every entry point below names the
 * reference item it replaces.  All structs are plain-old-data, all functions are `extern "C"`,
 * return an `int` status (0 = RT_OK, negative = error class) and never throw or abort across the
 * boundary; `rt_last_error()` returns the text of the last failure on the calling thread.
 *
 * Two groups of entry points live in the one shared library (ray_tracing_amd/_lib/librt_amd.so):
 *   rt_mesh_* / rt_scene_*   host-side scene preparation (the crate's mesh, Loop subdivision, STL
 *                            ingest and binned-SAH BVH builder, flattened into the SoA layout that
 *                            is uploaded once).  Pure CPU code, callable without a GPU.
 *   rt_create ... rt_render  the device side: hand-written HIP kernels for gfx950.
 *
 * All geometry and radiometry is IEEE binary32, evaluated in the reference's operation order with
 * no FMA contraction, so results are comparable bit-for-bit with the CPU restatement in oracle/.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes */
#define RT_OK 0
#define RT_ERR_INVALID (-1) /* bad argument / malformed scene (the reference would panic/assert) */
#define RT_ERR_HIP (-2)     /* a HIP runtime call failed; text in rt_last_error() */
#define RT_ERR_NOMEM (-3)
#define RT_ERR_IO (-4)    /* read_stl: io::Error (stl.rs:8) */
#define RT_ERR_STATE (-5) /* call order: no scene uploaded, ... */
#define RT_ERR_LIMIT (-6) /* a documented size limit was exceeded */

#define RT_MISS 0xFFFFFFFFu /* prim / obj value for "no hit" (Option::None) */

/* ---------------------------------------------------------------- enumerations */
enum rt_surface_kind {
    RT_SURF_MESH = 0,     /* Rc<BoundingVolumeHierarchy<TriangleRef>>  (triangle_mesh.rs:73) */
    RT_SURF_SPHERE = 1,   /* geometry::Sphere     (primitives.rs:26)  */
    RT_SURF_PLANE = 2,    /* geometry::Plane      (primitives.rs:215) */
    RT_SURF_TRIANGLE = 3, /* geometry::Triangle   (primitives.rs:111) */
    RT_SURF_POINT = 4     /* geometry::Point, never hit (primitives.rs:19-23) */
};
enum rt_material_kind {
    RT_MAT_DIFFUSE = 0,    /* bsdf.rs:43  */
    RT_MAT_DIELECTRIC = 1, /* bsdf.rs:88  (smooth only, as the reference asserts) */
    RT_MAT_CONDUCTOR = 2   /* bsdf.rs:142 (smooth only) */
};
enum rt_light_kind {
    RT_LIGHT_POINT = 0, /* lights.rs:30 */
    RT_LIGHT_SPOT = 1,  /* lights.rs:50 */
    RT_LIGHT_AREA = 2   /* lights.rs:77, sphere emitter */
};
enum rt_medium_kind {
    RT_MEDIUM_HOMOGENEOUS = 0, /* medium.rs:139 */
    RT_MEDIUM_CHECKERED = 1    /* testmedium.rs:12 */
};
#define RT_MEDIUM_NONE (-1) /* Option::None */
#define RT_MEDIUM_AUTO (-2) /* initial medium: let rt_upload_scene find it as Engine::new does */

/* ================================================================== scene DESCRIPTION
 * What a user of the crate writes in src/main.rs: meshes, objects with optional forward affine
 * transform, materials, lights, media and the camera constructor arguments.                      */

typedef struct rt_mesh rt_mesh; /* opaque TriangleMesh (triangle_mesh.rs:16-19) */

typedef struct rt_object_desc {
    uint32_t surface_kind;  /* enum rt_surface_kind */
    const rt_mesh* mesh;    /* RT_SURF_MESH: the mesh whose BVH is the surface */
    float surf[9];          /* SPHERE: center xyz, radius.  PLANE: point xyz, normal xyz (Plane::new
                               arguments, primitives.rs:222).  TRIANGLE: v0, v1, v2.  POINT: xyz. */
    uint32_t has_transform; /* Option<AffineTransform>: the FORWARD transform handed to Object::new */
    float matrix[9];        /* row-major 3x3 (math/matrix.rs:5) */
    float translation[3];
    uint32_t material;      /* index into materials */
    uint32_t has_medium_interface; /* Option<MediumInterface> (medium.rs:10) */
    int32_t medium_inside;  /* index into media or RT_MEDIUM_NONE */
    int32_t medium_outside;
} rt_object_desc;

typedef struct rt_material {
    uint32_t kind;  /* enum rt_material_kind */
    float a[3];     /* DIFFUSE reflectance; DIELECTRIC a[0] = refractive_index; CONDUCTOR n */
    float b[3];     /* CONDUCTOR extinction coefficient k */
} rt_material;

typedef struct rt_light {
    uint32_t kind;      /* enum rt_light_kind */
    float pos[3];       /* POINT/SPOT pos; AREA sphere center */
    float dir[3];       /* SPOT dir */
    float scalar;       /* SPOT cos_angle; AREA sphere radius */
    float intensity[3]; /* POINT/SPOT intensity; AREA emmited_radiance */
} rt_light;

typedef struct rt_medium {
    uint32_t kind;     /* enum rt_medium_kind */
    float sigma_a[3];  /* HOMOGENEOUS absorbtion_coefficient; CHECKERED [0] = sigma_a_white */
    float sigma_s[3];  /* HOMOGENEOUS scattering_coefficient; CHECKERED [0] = sigma_s_white */
    float emission[3]; /* HOMOGENEOUS emission; CHECKERED emission_white */
    float g;           /* Henyey-Greenstein asymmetry */
    float cell_size;   /* CHECKERED */
    float sigma_a_black, sigma_s_black;
    float emission_black[3];
} rt_medium;

typedef struct rt_camera_desc { /* Camera::new(origin, direction, aspect_ratio), camera.rs:14 */
    float origin[3];
    float direction[3];
    float aspect_ratio;
} rt_camera_desc;

typedef struct rt_scene_desc {
    uint32_t n_objects;
    const rt_object_desc* objects; /* Engine.objects, in vector order (path_integrator.rs:42) */
    uint32_t n_materials;
    const rt_material* materials;
    uint32_t n_lights;
    const rt_light* lights; /* Engine.lights */
    uint32_t n_media;
    const rt_medium* media;
    rt_camera_desc camera;
    int32_t initial_medium; /* RT_MEDIUM_AUTO (what Engine::new does), RT_MEDIUM_NONE or an index */
} rt_scene_desc;

/* ================================================================== FLATTENED scene
 * The compact SoA layout that crosses to the device exactly once.  A host that has its own BVH
 * builder (the Rust crate) fills this struct directly; rt_scene_build() fills it from a
 * description with the restated builder.                                                         */

typedef struct rt_bvh_flat { /* one BoundingVolumeHierarchy<TriangleRef> (bvh.rs:31-35) */
    uint32_t node_offset;    /* first node (the root) in the node_* arrays */
    uint32_t node_count;     /* <= 2*tri_count-1 (bvh.rs:62-66) */
    uint32_t tri_offset;     /* first triangle in the tri_* arrays */
    uint32_t tri_count;
    uint32_t max_depth;      /* depth of the deepest node, root = 0; sizes the traversal stack */
} rt_bvh_flat;

typedef struct rt_object_flat {
    uint32_t surface_kind;
    uint32_t bvh;           /* RT_SURF_MESH: index into bvhs */
    float surf[9];          /* SPHERE center,radius. PLANE normal xyz (unit), d (primitives.rs:216-217).
                               TRIANGLE v0,v1,v2 */
    uint32_t has_transform; /* object.rs:11 */
    float inv_matrix[9];    /* the STORED (inverse) transform, object.rs:25 */
    float inv_translation[3];
    uint32_t material;
    uint32_t has_medium_interface;
    int32_t medium_inside;
    int32_t medium_outside;
} rt_object_flat;

typedef struct rt_camera_flat { /* the four fields Camera::ray reads (camera.rs:60-63) */
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
} rt_camera_flat;

typedef struct rt_scene_flat {
    /* BVH nodes of all meshes, each mesh's nodes in the reference's order (root first, the two
       children of node k adjacent at node_index[k], node_index[k]+1, both LOCAL to the mesh). */
    uint32_t n_nodes;
    const float* node_upper;       /* [n_nodes][3]  Node.bounding_box.upper (primitives.rs:251) */
    const float* node_lower;       /* [n_nodes][3]  Node.bounding_box.lower */
    const uint32_t* node_index;    /* [n_nodes]     Node.index   (bvh.rs:16) */
    const uint32_t* node_num_prim; /* [n_nodes]     Node.num_prim(bvh.rs:17); leaf iff > 0 */
    /* Triangles in LEAF ORDER: position i of a mesh holds primitives[indices[i]] (bvh.rs:84-86),
       so a leaf {index, num_prim} owns tri_offset+index .. +num_prim contiguously. */
    uint32_t n_tris;
    const float* tri_v0;         /* [n_tris][3] */
    const float* tri_v1;         /* [n_tris][3] */
    const float* tri_v2;         /* [n_tris][3] */
    const uint32_t* tri_prim_id; /* [n_tris] original triangle index in its mesh (= indices[i]) */
    uint32_t n_bvhs;
    const rt_bvh_flat* bvhs;
    uint32_t n_objects;
    const rt_object_flat* objects;
    uint32_t n_materials;
    const rt_material* materials;
    uint32_t n_lights;
    const rt_light* lights;
    uint32_t n_media;
    const rt_medium* media;
    rt_camera_flat camera;
    int32_t initial_medium;
} rt_scene_flat;

/* ================================================================== host-side scene preparation */

/* TriangleMesh { vertices, triangles } (triangle_mesh.rs:16-19).  Copies the arrays. */
int rt_mesh_create(const float* vertices, uint32_t n_vertices, const uint32_t* triangles,
                   uint32_t n_triangles, rt_mesh** out);
/* render::stl::read_stl (stl.rs:8): binary STL + bit-exact vertex unification (stl.rs:114-143). */
int rt_mesh_read_stl(const char* path, rt_mesh** out);
/* geometry::subdivision_surface::loop_subdivide (subdivision_surface.rs:18-25):
   `divisions` Loop splits, then push_to_limit_surface. */
int rt_mesh_loop_subdivide(const rt_mesh* mesh, uint32_t divisions, rt_mesh** out);
/* TriangleMesh::apply_transform (triangle_mesh.rs:83-87): v <- M v + t for every vertex. */
int rt_mesh_apply_transform(rt_mesh* mesh, const float matrix[9], const float translation[3]);
/* TriangleMesh::volume_centroid (triangle_mesh.rs:103-118), used by main.rs to aim cameras. */
int rt_mesh_volume_centroid(const rt_mesh* mesh, float out[3]);
int rt_mesh_counts(const rt_mesh* mesh, uint32_t* n_vertices, uint32_t* n_triangles);
int rt_mesh_copy_out(const rt_mesh* mesh, float* vertices, uint32_t* triangles);
void rt_mesh_destroy(rt_mesh* mesh);

typedef struct rt_scene rt_scene; /* owns the flattened arrays */

/* TriangleMesh::build_bvh for every distinct mesh (triangle_mesh.rs:73-81 -> bvh.rs:60-241, the
   binned-SAH builder restated node for node), Object::new (object.rs:17-29, stores the inverse
   transform), Plane::new (primitives.rs:222-226) and Camera::new (camera.rs:14-53); then flattens. */
int rt_scene_build(const rt_scene_desc* desc, rt_scene** out);
/* Borrowed view of the flattened arrays; valid until rt_scene_destroy. */
int rt_scene_view(const rt_scene* scene, rt_scene_flat* out);
void rt_scene_destroy(rt_scene* scene);

/* write_pixel (path_integrator.rs:19-27) + the NaN guard and /spp of Engine::render (:309-312):
   rgb_sum[j*w+i][3] (row j = the reference's v index, 0 = bottom) -> 8-bit rows top to bottom. */
int rt_tonemap_u8(const float* rgb_sum, uint32_t width, uint32_t height, uint32_t samples_per_pixel,
                  uint8_t* rgb8_out);
/* The P3 text PPM Engine::render streams to stdout (path_integrator.rs:293,:312). */
int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* ================================================================== device side */

typedef struct rt_ctx rt_ctx; /* one context per GPU; not thread-safe per context */

/* Opens `device` (hipSetDevice) and creates the stream all work of this context runs on. */
int rt_create(int device, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);

/* Engine::new (path_integrator.rs:50-63): copies the flattened scene to HBM once (re-laid out as
   64-byte child pairs and 48-byte triangles) and, for RT_MEDIUM_AUTO, runs set_initial_medium
   (:65-85) with the device intersector. */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_flat* scene);
/* Engine.initial_medium after upload (index or RT_MEDIUM_NONE). */
int rt_initial_medium(const rt_ctx* ctx, int32_t* out);

/* SoA ray batch; pointers may be host or device memory (detected), n entries each. */
typedef struct rt_rays {
    const float *ox, *oy, *oz; /* Ray.origin    (ray.rs:5) */
    const float *dx, *dy, *dz; /* Ray.direction (ray.rs:6) */
} rt_rays;

/* Closest-hit record, SoA; every pointer is optional (NULL = not wanted) except t. Host or device. */
typedef struct rt_hits {
    float* t;            /* SurfaceIntersection.t, +inf for a miss */
    uint32_t* prim;      /* original triangle index in its mesh (0 for analytic surfaces), RT_MISS */
    uint32_t* obj;       /* index of the Object that was hit, RT_MISS */
    float *nx, *ny, *nz; /* SurfaceIntersection.normal (world space) */
    uint8_t* front_face; /* SurfaceIntersection.front_face */
    uint32_t* n_int;     /* visit counter: internal BVH nodes popped (bvh.rs:271-274) */
    uint32_t* n_tri;     /* visit counter: triangles tested (bvh.rs:256-257) */
} rt_hits;

/* Batch form of Engine::hit (path_integrator.rs:88-97): linear scan of the objects with shrinking
   t_max -> Object::hit (object.rs:32-68) -> Surface::hit; meshes go through
   BoundingVolumeHierarchy::traverse_bvh(Nearest) (bvh.rs:243-300). */
int rt_trace_closest(rt_ctx* ctx, const rt_rays* rays, size_t n, float t_min, float t_max,
                     const rt_hits* hits);
/* Batch form of Engine::hit_bool (path_integrator.rs:100-104) -> traverse_bvh(Any).
   occluded[i] = 1 if any object is hit in (t_min, t_max).  n_int/n_tri optional visit counters. */
int rt_trace_any(rt_ctx* ctx, const rt_rays* rays, size_t n, float t_min, float t_max,
                 uint8_t* occluded, uint32_t* n_int, uint32_t* n_tri);

/* Allocation-free form of rt_trace_closest.  NO counterpart in the reference: it exists for callers that cannot
   call hipMalloc -- tight serving loops, harnesses that forbid allocation.  (Not capturable into a HIP graph as it
   stands: it launches on the context's own stream and synchronises it before it returns.)  Same
   results as rt_trace_closest, bit for bit, but it calls no hipMalloc / hipFree / allocating routine: the
   launch's scratch (work counters + traversal-stack overflow strips) lives in a static device array of
   24.016 MiB that every GPU the library runs on holds until the process ends.  Limits: every ray / hit array
   must be DEVICE memory (no staging), n <= RT_NOALLOC_MAX_RAYS, and the scene's deepest BVH must not exceed
   RT_NOALLOC_MAX_STACK_DEPTH levels (RT_ERR_LIMIT otherwise).  All calls on one GPU share the static array:
   calls that overlap on the same GPU (two contexts, two threads) race; serialise them. */
#define RT_NOALLOC_MAX_RAYS 0xFFFFFFF0ull
#define RT_NOALLOC_MAX_STACK_DEPTH 32
int rt_trace_closest_noalloc(rt_ctx* ctx, const rt_rays* rays, size_t n, float t_min, float t_max,
                             const rt_hits* hits);

typedef struct rt_stats {
    uint64_t rays_closest;  /* Engine::hit calls issued (extend kernel lanes) */
    uint64_t rays_shadow;   /* Engine::hit_bool calls the reference's algorithm issues (= the oracle's count); of these,
                               rays_shadow_skipped were not traced (below) */
    uint64_t paths;         /* camera paths started = width*height*spp_count */
    uint64_t nodes_visited; /* sum of n_int over the closest-hit rays (only with RT_RENDER_COUNT_VISITS) */
    uint64_t tris_tested;   /* sum of n_tri over the closest-hit rays (only with RT_RENDER_COUNT_VISITS) */
    uint64_t nodes_visited_shadow; /* the same two counters for the shadow rays */
    uint64_t tris_tested_shadow;
    uint32_t iterations;    /* wavefront iterations run (max over batches) */
    uint32_t batches;
    float ms_total;         /* device time of the whole call (HIP events on the context stream) */
    float ms_generate, ms_extend, ms_shade, ms_shadow, ms_compact;
    uint64_t launches_extend; /* number of extend launches (for per-launch averages) */
    uint64_t launches_shadow;
    /* Share of the totals above traced outside the per-bounce extend / shadow launches: late
       bounces trace their shadow rays and the next bounce's rays in one launch, and once few paths are
       alive ONE launch carries each of them through all its remaining bounces (paths_tail). */
    uint64_t paths_tail;
    uint64_t rays_closest_tail, rays_shadow_tail;
    uint64_t nodes_visited_tail, tris_tested_tail; /* closest-hit rays of the tail kernel */
    uint64_t launches_tail;
    float ms_tail;            /* always filled, like ms_extend */
    uint32_t lean_kernels;    /* 1: the scene has no analytic sphere/triangle objects and the lean traversal kernels (k_*_lean) ran */
    uint64_t nodes_visited_shadow_tail, tris_tested_shadow_tail; /* shadow rays of the mixed / tail launches */
    /* Shadow rays the reference traces (path_integrator.rs:219) whose light term is exactly zero or never added
       (p_light = 0, :230): whatever they hit, no bit of the radiance depends on it, so rt_render does not trace
       them.  Counted in rays_shadow, not traced.  Always 0 with RT_RENDER_COUNT_VISITS (every ray is traced so
       that the visit counters equal the oracle's). */
    uint64_t rays_shadow_skipped;
} rt_stats;

#define RT_RENDER_COUNT_VISITS 1u /* fill nodes_visited / tris_tested with the REFERENCE's visits: no root-box
                                     skip, no top-level cull -- equal to the oracle's counters (slower variant) */
#define RT_RENDER_COUNT_FETCHES 8u /* fill them with the visits the TIMED kernels perform (root-box skip and
                                      top-level cull active): the fetches the roofline figure prices */
#define RT_RENDER_NO_TAIL 4u      /* keep one launch per stage and bounce to the end (no tail kernel) */
#define RT_RENDER_TIME_STAGES 2u  /* fill every per-stage ms_* field (adds event records); ms_extend is always filled */

/* Engine::render (path_integrator.rs:278-318) minus the PPM text: for every pixel (i, j) adds
   ray_colour (:138-276) of samples [spp_first, spp_first+spp_count) IN SAMPLE ORDER and stores the
   sum in rgb_sum[(j*width+i)*3 .. +3] (j = the reference's v index, row 0 = bottom).  rgb_sum may
   be host or device memory.  Random numbers: Philox4x32-10 keyed by seed, counter
   (pixel=j*width+i, global sample index, bounce, draw/4), word draw%4 -- see DESIGN.md.  stats may be NULL. */
int rt_render(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp_first, uint32_t spp_count,
              uint32_t depth, uint64_t seed, uint32_t flags, float* rgb_sum, rt_stats* stats);

/* ---- multi-GPU (no counterpart in the reference, which is one thread: path_integrator.rs:286-308).
   One process per GPU.  Rank r renders global samples [first_r, first_r + count_r) of every pixel into its own
   device accumulator (rt_render's spp_first / spp_count; the random numbers are keyed by the GLOBAL sample
   index, so the union over ranks is the single-GPU frame) and the accumulators are summed with one RCCL
   reduce over xGMI.  RCCL (librccl.so.1) is bound at run time; without it these calls return RT_ERR_STATE. */
typedef struct rt_comm rt_comm;
#define RT_COMM_ID_BYTES 128
/* ncclGetUniqueId: called on ONE rank; the host hands the bytes to every other rank (file, MPI, ...). */
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
/* ncclCommInitRank on ctx's device; collective over all world_size ranks. */
int rt_comm_create(rt_ctx* ctx, const uint8_t id[RT_COMM_ID_BYTES], int world_size, int rank, rt_comm** out);
/* Waits for ctx's stream, then ncclCommDestroy.  Call it BEFORE rt_destroy(ctx): the communicator keeps the pointer. */
void rt_comm_destroy(rt_comm* comm);
/* Sum of the ranks' rgb_sum[height][width][3] (DEVICE memory, in place) onto rank `root` (ncclReduce), or onto
   every rank when root < 0 (ncclAllReduce).  Enqueued on the context's stream, i.e. behind the rt_render call
   that filled rgb_sum; returns when the result is complete.  24.9 MB at 1080p, 99.5 MB at 4K. */
int rt_reduce_accum(rt_comm* comm, float* rgb_sum, uint32_t width, uint32_t height, int root);

/* ---- frames in flight (no counterpart in the reference: Engine::render is one synchronous loop,
   path_integrator.rs:278-318).  The late bounces of a frame are a few long walks that leave most of the GPU idle; a
   pipeline renders up to `frames_in_flight` DIFFERENT frames at once on one GPU -- one context (stream, scene copy,
   path state) and one host thread each, frames dealt round-robin, every frame by the unchanged rt_render (same bits
   as a lone call) -- so that the large launches of one frame fill what the late launches of another leave empty.
   Throughput, not latency: a frame in a full pipeline takes longer than alone.  One host thread drives a pipeline. */
typedef struct rt_pipeline rt_pipeline;
#define RT_PIPELINE_MAX_FRAMES 8
#define RT_PIPELINE_NO_REDUCE (-2) /* reduce_root: render only */
int rt_pipeline_create(int device, uint32_t frames_in_flight, rt_pipeline** out);
/* Renders what is still queued, then releases the communicators and the contexts. */
void rt_pipeline_destroy(rt_pipeline* pipe);
uint32_t rt_pipeline_frames(const rt_pipeline* pipe);
/* Streams of one process share a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and work of two streams on ONE
   hardware queue runs in order, not side by side.  rt_pipeline_create tests its contexts' streams against each other
   (a 2 ms spin kernel on one, an empty kernel on the other) and replaces a stream that had to wait.  This is the number
   of slots seen to run beside all slots before them (slot 0 included): frames_in_flight when every frame in flight has a
   hardware queue of its own, less when the process has more busy streams than the GPU has queues. */
uint32_t rt_pipeline_separate_queues(const rt_pipeline* pipe);
/* The context of one slot, borrowed: for settings (rt_set_visit_count_mode, ...) and for plain rt_render calls
   while no frame is in flight. */
rt_ctx* rt_pipeline_context(rt_pipeline* pipe, uint32_t slot);
/* rt_upload_scene on every context (one scene copy each); shares 80 % of the free HBM among the contexts' path
   states (rt_set_max_paths_in_flight).  Only while no frame is in flight (RT_ERR_STATE otherwise). */
int rt_pipeline_upload_scene(rt_pipeline* pipe, const rt_scene_flat* scene);
/* Multi-GPU: one communicator per slot from ids[frames_in_flight][RT_COMM_ID_BYTES] (rt_comm_unique_id on one rank,
   once per slot), created slot by slot -- collective, every rank calls it with the same ids. */
int rt_pipeline_comm_create(rt_pipeline* pipe, const uint8_t* ids, int world_size, int rank);
/* Queues one rt_render call (same arguments; rgb_sum must stay valid and untouched until the frame has been waited
   for, and frames in flight need distinct buffers) and returns its ticket at once.  reduce_root >= 0 / -1: the frame
   is followed by rt_reduce_accum onto that rank / onto every rank (rgb_sum must be device memory); the reduces of
   successive frames are issued in submission order, one at a time, whatever order the frames finish in -- every
   rank must submit the same frames in the same order.  RT_PIPELINE_NO_REDUCE: none. */
int rt_pipeline_submit(rt_pipeline* pipe, uint32_t width, uint32_t height, uint32_t spp_first, uint32_t spp_count,
                       uint32_t depth, uint64_t seed, uint32_t flags, float* rgb_sum, int reduce_root, uint64_t* ticket);
/* Blocks until the frame (and its reduce) is complete; returns ITS status, with the text in rt_last_error() on the
   calling thread, and its statistics (the ms_* fields are stream times that include the other frames' kernels).
   Every ticket is waited for exactly once. */
int rt_pipeline_wait(rt_pipeline* pipe, uint64_t ticket, rt_stats* stats);

/* The same pipeline with the two per-frame steps supplied by the caller (no counterpart in the reference): `render`
   runs on the slot's worker thread with the slot's context -- any sequence of calls on it, e.g. rt_ray_colour for
   frames of caller rays, or rt_render followed by the host's own kernels on rt_stream(ctx) -- and `reduce`, if the
   frame was submitted with reduce != 0, afterwards, in submission order and one frame at a time across the slots (the
   guarantee rt_pipeline_submit's reduces have; it is what lets every rank of a multi-GPU job pair its collectives).
   Both return RT_OK or an RT_ERR_* status, which rt_pipeline_wait hands back.  device < 0: no contexts (ctx is NULL in
   the hooks), the frames are the caller's own host work -- used by the CPU test of the ordering over gloo
   (tests/test_pipeline_order_cpu.py). */
typedef struct rt_pipeline_hooks {
    int (*render)(void* user, rt_ctx* ctx, uint32_t slot, uint64_t ticket);
    int (*reduce)(void* user, rt_ctx* ctx, uint32_t slot, uint64_t ticket); /* may be NULL */
    void* user;
} rt_pipeline_hooks;
int rt_pipeline_create_hooked(int device, uint32_t frames_in_flight, const rt_pipeline_hooks* hooks, rt_pipeline** out);
int rt_pipeline_submit_hooked(rt_pipeline* pipe, int reduce, uint64_t* ticket);

/* Engine::ray_colour (path_integrator.rs:138-276, the crate's public per-ray entry) in batch form: rgb_sum[3k..] =
   sum over the samples [spp_first, spp_first + spp_count) of ray_colour(ray k, depth).  Ray k is taken as it is (the
   reference's callers pass Camera::ray's normalised direction); the random numbers of (ray k, sample s) are those
   of pixel index k and sample s from bounce 1 on -- rendering the rays Camera::ray produces
   for the jittered pixel positions gives rt_render's image.  Same pipeline, flags and statistics as rt_render;
   rays and rgb_sum (3 n floats) may be host or device memory. */
int rt_ray_colour(rt_ctx* ctx, const rt_rays* rays, size_t n, uint32_t spp_first, uint32_t spp_count, uint32_t depth,
                  uint64_t seed, uint32_t flags, float* rgb_sum, rt_stats* stats);

/* Upper bound on camera paths kept in flight by rt_render (default: sized from free HBM). */
int rt_set_max_paths_in_flight(rt_ctx* ctx, uint64_t max_paths);

/* What the optional n_int / n_tri outputs of rt_trace_closest / rt_trace_any count.  RT_COUNT_REFERENCE
   (default): the reference's visits (bvh.rs:256-257, :271-274), root of every mesh included, equal to the
   oracle's per ray.  RT_COUNT_FETCHES: the visits the non-counting kernels perform (a mesh whose root box or
   world bounds the ray misses costs none).  Results (t, prim, obj, ...) are the same bits in every mode.
   No counterpart in the reference; used by bench.py for the roofline figure. */
#define RT_COUNT_REFERENCE 0u
#define RT_COUNT_FETCHES 1u
int rt_set_visit_count_mode(rt_ctx* ctx, uint32_t mode);

/* Device time (ms) of the kernel launched by the most recent rt_trace_closest / rt_trace_any call:
   HIP events recorded on the context stream directly before and after that one launch (staging
   copies excluded).  No counterpart in the reference; used by bench.py for the roofline figure. */
int rt_last_kernel_ms(rt_ctx* ctx, float* ms);

/* The HIP stream the context launches on (a hipStream_t), for callers that time with HIP events. */
void* rt_stream(rt_ctx* ctx);

const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
