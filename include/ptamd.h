/* ptamd.h -- C ABI of the MI355X (gfx950) ray-queue render path.
 *
 * Drop-in boundary for ONE path of mathijs727/OpenCL-Path-Tracer: the ray-queue render loop
 *   generatePrimaryRays -> intersectWalk -> shade -> intersectShadows (+ updateKernelData, accumulate)
 * i.e. everything the reference's RayTracer does through src/opencl + assets/cl.  Each entry point
 * names the reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions: every call returns 0 on success or a negative pt_status; the message of the last
 * failure on a context is available from pt_last_error().  Nothing here ever exits the process
 * (the reference's checkClErr does, src/opencl/cl_helpers.cpp:21-31).  All scene inputs use the byte
 * layouts the reference uploads to its device buffers, so a maintainer
 * passes the host vectors RayTracer already owns (src/raytracer.h:86-91) unchanged; the library
 * converts them to its own SoA / packed layouts in HBM during pt_upload_*.
 * A context is bound to one GPU and is not re-entrant; use one context per device.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID = -1, /* bad argument / inconsistent scene */
    PT_ERR_HIP = -2, /* a HIP runtime call failed */
    PT_ERR_STATE = -3, /* call order (e.g. render before upload) */
    PT_ERR_UNSUPPORTED = -4
} pt_status;

/* ---- scene arrays: byte-for-byte the reference's device structs ---------------------- */

/* VertexData, assets/cl/shapes.cl:13-18 == VertexSceneData, src/vertices.h:11-16 (48 B) */
typedef struct {
    float vertex[4]; /* xyz, w unused */
    float normal[4];
    float texCoord[2];
    float _pad[2];
} pt_vertex;

/* TriangleData, assets/cl/shapes.cl:7-11 == TriangleSceneData, src/vertices.h:6-9 (16 B) */
typedef struct {
    uint32_t indices[3]; /* GLOBAL vertex indices (rebased, src/raytracer.cpp:254-258) */
    uint32_t materialIndex; /* GLOBAL material index */
} pt_triangle;

/* Material, assets/cl/material.cl:3-51 == src/model/material.h:8-61 (48 B tagged union) */
typedef enum { PT_MAT_DIFFUSE = 0, PT_MAT_PBR = 1, PT_MAT_REFRACTIVE = 2, PT_MAT_BASIC_REFRACTIVE = 3, PT_MAT_EMISSIVE = 4 } pt_material_type;
typedef struct {
    union {
        struct { float diffuseColour[4]; int32_t textureId; } diffuse; /* textureId -1: untextured */
        struct { float baseColour[4]; float smoothness; float f0NonMetal; uint8_t metallic; } pbr;
        struct { float absorption[4]; float smoothness; float refractiveIndex; } refractive;
        struct { float absorption[4]; float refractiveIndex; } basicRefractive;
        struct { float emissiveColour[4]; } emissive;
        uint8_t _raw[32];
    } u;
    int32_t type; /* pt_material_type */
    uint8_t _pad[12];
} pt_material;

/* EmissiveTriangle, assets/cl/light.cl:5-9 == src/vertices.h:18-21 (96 B), WORLD space */
typedef struct {
    float vertices[3][4];
    pt_material material;
} pt_emissive_triangle;

/* SubBvhNode, assets/cl/bvh.cl:5-16 == SubBVHNode, src/bvh/bvh_nodes.h:35-49 (48 B).
 * Leaf iff triangleCount != 0; an inner node's children are (leftChildIndex, leftChildIndex+1). */
typedef struct {
    float min[4];
    float max[4];
    uint32_t leftChildOrFirstTriangle; /* GLOBAL (rebased, src/raytracer.cpp:262-269) */
    uint32_t triangleCount;
    uint32_t _pad[2];
} pt_sub_bvh_node;

/* TopBvhNode, assets/cl/bvh.cl:18-34 == TopBVHNode, src/bvh/bvh_nodes.h:20-33 (112 B).
 * Leaf: a = root of the instance's sub-BVH (global node index), invTransform = inverse(world),
 * column-major.  Inner: a = left child, b = right child (top-node indices). */
typedef struct {
    float min[4];
    float max[4];
    float invTransform[16];
    uint32_t a;
    uint32_t b;
    uint32_t isLeaf;
    uint32_t _pad;
} pt_top_bvh_node;

/* Camera, assets/cl/camera.cl:7-26 == CameraData, src/camera.h:7-27 (128 B) */
typedef struct {
    float eyePoint[4];
    float screenPoint[4]; /* top-left corner of the virtual screen */
    float u[4];
    float v[4];
    float uNormalized[4];
    float vNormalized[4];
    float focalDistance;
    float apertureRadius;
    float relativeAperture; /* f-stops */
    float shutterTime;
    float ISO;
    uint8_t thinLensEnabled;
    uint8_t _pad[11];
} pt_camera;

/* ---- configuration -------------------------------------------------------------------- */

typedef enum {
    /* production: counter-based PRNG keyed by (pixel, sample, depth, dimension, seed); wave-ballot
     * compaction in arbitrary wave order.  Images are bit-reproducible run to run. */
    PT_RNG_COUNTER = 0,
    /* parity: clRNG LFSR113 streams bound to queue slots exactly like the reference
     * (kernel.cl:43,246; src/raytracer.cpp:739-751) with stable (slot-ordered) compaction, so a
     * render follows the reference's kernels run in work-item order (oracle/_ref). */
    PT_RNG_LFSR113_PARITY = 1
} pt_rng_mode;

typedef struct {
    uint32_t width, height; /* full image size (all ranks) */
    uint32_t max_active_rays; /* queue capacity; 0 = one slot per owned pixel (no refill).
                                 Reference: MAX_ACTIVE_RAYS = 1280*720, src/raytracer.cpp:40 */
    uint32_t max_bounces; /* 0 = 4 (MAX_ITERATIONS, assets/cl/kernel.cl:4) */
    uint32_t rng_mode; /* pt_rng_mode */
    uint32_t seed; /* PT_RNG_COUNTER only */
    int32_t device; /* HIP device ordinal */
    uint32_t flags; /* PT_FLAG_* */
    uint32_t samples_in_flight; /* samples of one pixel traced concurrently, each into its own accumulator
                                   plane (folded at the end of pt_render); 0 = auto (~32M path segments per launch), max 4096.
                                   Rounded DOWN to a multiple of 256, below 256 to a power of two (only such batches keep
                                   the samples of a pixel next to each other in the queues).
                                   Only with max_active_rays == 0 and PT_RNG_COUNTER. */
    /* Round 6: queues SMALLER than a batch (0 = 1: one slot per entry, as the reference allocates, src/raytracer.cpp:760-787).  A batch of B entries needs B hit
     * records and B camera-ray directions, but its second extension queue only holds the paths that go on after the first hit and its shadow queue the shadow
     * rays of the first hits -- 26 % and 58 % of B on BASELINE config 4.  With fractions f_ext, f_shadow in (0, 1) those queues hold f x (owned pixels x
     * samples_in_flight) entries: 36 + 80 f_ext + 48 f_shadow bytes per entry instead of 164 (pinhole; a thin lens keeps its camera-ray origins: 68 + 48 f_ext +
     * 48 f_shadow).  pt_render then sizes every batch so that what its first pass emits fits -- from the counts of earlier batches of the same camera, scene
     * and tiles; a short probe batch first -- i.e. a scene that emits more than the fractions allow renders in smaller batches, never wrongly; a guess that
     * turns out wrong all the same is cut at the queue's end on the device and REPORTED: pt_synchronize and the image reads fail (PT_ERR_STATE).
     * Fixed schedule only (max_active_rays == 0, PT_RNG_COUNTER, samples_in_flight >= 16). */
    float ext_queue_fraction;
    float shadow_queue_fraction;
} pt_config;

#define PT_FLAG_ROWMAJOR_PIXELS 1u /* issue pixels in row-major order (reference order); default is 8x8 blocks */
#define PT_FLAG_NO_BAKED_INSTANCES 2u /* every instance is ENTERED at traversal like in the reference (scene.cl:116-139): no world-space copies at all */
#define PT_FLAG_TWO_LEVEL_ONLY 4u /* copy only single-leaf instances (quads, lights) to world space, not whole meshes */
#define PT_FLAG_NO_PACKETS 8u /* never use the packet traversal kernel (primary rays then go through the per-ray kernel) */
#define PT_FLAG_PACKET_INTERSECT 16u /* pt_intersect (test hook) uses the packet kernel where the scene allows it */
/* shade runs neeMisShading (assets/cl/shading.cl:35-349: NEE + BSDF sampling combined by the balance heuristic) instead of
 * neeIsShading (:356-623), the integrator the reference compiles in.  The reference reaches its MIS code only under
 * #define COMPARE_SHADING (kernel.cl:6); one uninitialised read in it is fixed here. */
#define PT_FLAG_MATERIAL_BINS 512u /* scenes whose surfaces are of several material types: k_shade walks every 512-entry tile in material
                                     order instead of queue order (measured SLOWER on MI355X -- the kernel is bound by its gathers, not by
                                     divergent BSDF code: DESIGN.md section 6 -- hence opt-in) */
#define PT_FLAG_QUEUE_PRIMARY_RAYS 256u /* always write the primary rays to the queue (k_gen), also where the packet kernel could
                                          regenerate them from the entry index (diagnostics; the image is the same) */
/* 1024u, 2048u: retired (rounds 4-5: an opt-in shared descent ahead of the per-ray traversal, measured no faster; EXPERIMENTS.md) -- ignored */
#define PT_FLAG_PARKED_INSTANCES 4096u /* every instance that is entered takes the general route (ray transformed into the instance's space, lane parked for entry and exit:
                                         rounds 2-4).  Default since round 5: instances whose transform is a translation + uniform scale are walked by the per-ray
                                         kernels without parking (entry nodes, the ray taken into the instance's space on the fly; csrc/pt_trace.h); rotated /
                                         non-uniformly scaled ones keep the general route either way */
#define PT_FLAG_TEAM_INTERSECT 8192u /* pt_intersect (test hook) uses the team kernel (four lanes per ray, csrc/pt_team.h) where the scene allows it */
#define PT_FLAG_INTEGRATOR_MIS 32u
/* exactly the reference's COMPARE_SHADING build (kernel.cl:48-51,248-265; raytracer.cpp:464-495): neeMisShading for the pixels of
 * the left half of the image, neeIsShading for the right half, both halves showing the left half's view -- two estimators of one
 * image side by side, whose mean luminances must agree */
#define PT_FLAG_COMPARE_SHADING 64u
/* next event estimation picks its light with probability proportional to the solid angle each emissive triangle subtends at the
 * shading point (weightedRandomPointOnLight, shading_helper.cl:216-259) instead of uniformly (randomPointOnLight, :261-278) */
#define PT_FLAG_SOLID_ANGLE_LIGHTS 128u

typedef struct { uint32_t x0, y0, x1, y1; } pt_rect; /* [x0,x1) x [y0,y1) */

typedef struct {
    uint64_t rays_extension; /* traceRay calls from intersectWalk on live entries */
    uint64_t rays_shadow; /* traceRay calls from intersectShadows on live entries */
    uint64_t rays_generated; /* primary rays */
    uint64_t shade_hits; /* shade invocations on a hit */
    uint64_t deposits; /* accumulator updates */
    uint64_t samples; /* samples per pixel rendered since last reset */
    double ms_last_render; /* device time of the last pt_render (hipEvent, whole call) */
    double ms_intersect, ms_shade, ms_shadow, ms_gen; /* per-kernel-family device ms of the last pt_render
                                                        (only when PT_PROFILE_KERNELS was requested) */
    uint64_t packet_launches; /* launches of the packet traversal kernels (first pass of a batch with >= 16 samples of a pixel next to each other) */
    double ms_packet; /* the part of ms_intersect spent in the packet traversal kernel */
    uint64_t deposits_shadow; /* the part of `deposits` made by the any-hit traversal (unoccluded shadow rays,
                                 kernel.cl:132-135); the rest are emissive hits and sky misses in shade */
    uint64_t gen_launches; /* launches of the primary-ray kernel (generatePrimaryRays); 0 where the packet traversal kernel generates the camera rays itself */
    uint64_t bundle_launches; /* of packet_launches: those that walked the tree once per bundle of several packets (camera rays of a pinhole, generated in the kernel) */
    uint32_t stack_need; /* worst-case traversal stack entries of the active scene state (packet kernels need <= 64) */
    uint32_t folded_instances; /* instances of the active scene state that the per-ray kernels walk without parking (translation + uniform scale, not copied to world space) */
    uint64_t team_launches; /* traversal launches served by the team kernel (four lanes per ray: launches that do not fill the machine, csrc/pt_team.h) */
    uint32_t entered_instances; /* instances of the active scene state that are entered at traversal (scene.cl:116-139) rather than copied to world space */
    uint32_t batch_samples; /* samples per pixel of the last batch pt_render cut (samples_in_flight, or fewer where the queue fractions of pt_config made it) */
    float first_pass_ext_ratio, first_pass_shadow_ratio; /* the largest (rays emitted by a batch's first pass) / (entries of the batch) seen for the current camera, scene
                                                            state and tiling -- extension rays, shadow rays; 0: not measured (queues as large as the batch) */
    uint32_t probe_batches; /* short batches rendered to learn what a batch's first pass emits (queue fractions of pt_config; once per camera / scene state / tiling) */
    uint32_t general_route; /* 1: the per-ray kernels enter instances of ANY transform as leaf-kind steps, nothing parked (csrc/pt_trace.h, LEVELS 2: scenes with a rotated /
                               non-uniformly scaled instance, or with more instances than the fold table holds) */
} pt_stats;

typedef struct pt_ctx pt_ctx;

/* ---- lifetime -- replaces CLContext + RayTracer ctor/dtor (src/opencl/context.cpp, src/raytracer.cpp:63-86) */
int pt_create(const pt_config* cfg, pt_ctx** out);
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(const pt_ctx* ctx); /* ctx may be NULL: error of a failed pt_create */
/* Run everything on a caller-owned HIP stream; NULL = a non-blocking stream owned by the context (the default).
 * NOTE: the handle of the legacy default stream IS NULL (torch.cuda.current_stream().cuda_stream == 0 unless a
 * torch.cuda.Stream is current), so "torch's default stream" cannot be selected this way: work enqueued there by
 * others -- a collective, a tensor copy -- is NOT ordered after the render.  Pass an explicit stream and issue the
 * dependent work on it, or call pt_synchronize first. */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);

/* ---- uploads -- replace the enqueueWriteBuffer calls of src/raytracer.cpp:273-282 and :556-578 */
int pt_upload_static(pt_ctx* ctx, const pt_vertex* verts, uint32_t n_verts, const pt_triangle* tris, uint32_t n_tris,
    const pt_material* mats, uint32_t n_mats, const pt_sub_bvh_node* nodes, uint32_t n_nodes);
int pt_upload_dynamic(pt_ctx* ctx, const pt_emissive_triangle* lights, uint32_t n_lights,
    const pt_top_bvh_node* top_nodes, uint32_t n_top, uint32_t top_root);
/* A REBUILT scene as a frame-loop citizen -- the other branch of MeshSequence::buildBvh (src/model/mesh_sequence.cpp:89-96: a new tree per frame
 * instead of a refit), whose arrays transferDynamicData re-uploads every tick (src/raytracer.cpp:510-568).  Same arguments as pt_upload_static;
 * the scene is converted into the context's SECOND static set and copied to the device WITHOUT synchronising the render stream: frames enqueued
 * so far, and any enqueued before the flip, keep rendering the old trees.  Follow with pt_upload_dynamic_async -- lights and top level of the
 * NEW scene (its leaves name the new sub-BVH roots) -- and pt_frame_tick, which adopts both.  (Before anything renders it is pt_upload_static.)
 * The host converts the topology; the node boxes, quantised planes and triangle records are made on the device, on the copy stream, from the arrays as
 * handed in (their bytes are those pt_upload_static makes on the host). */
int pt_upload_static_async(pt_ctx* ctx, const pt_vertex* verts, uint32_t n_verts, const pt_triangle* tris, uint32_t n_tris,
    const pt_material* mats, uint32_t n_mats, const pt_sub_bvh_node* nodes, uint32_t n_nodes);
/* transferDynamicData + frameTick as the reference runs them (src/raytracer.cpp:183-189,497-595): the dynamic part of the scene
 * is double-buffered on the device.  pt_upload_dynamic_async converts the next state on the host ("Lot of CPU work", :185) and
 * copies it into the INACTIVE buffers on a copy stream of its own -- the GPU keeps rendering the active state meanwhile, nothing
 * waits on the host; pt_frame_tick makes the render stream wait for that copy (the barrier of :593) and flips m_activeBuffer.
 * pt_upload_dynamic is the two in a row.
 * Limits (PT_ERR_UNSUPPORTED, with the way out in pt_last_error): the packed nodes and the triangle records of a state -- the world-space copies
 * of instances included -- are addressed by 32-bit byte offsets on the device: at most 4 GB of either (64 M nodes, 89 M triangle records); a
 * scene whose copies would pass that, or the library's 2 GB copy budget, is traversed with its instances entered (PT_FLAG_NO_BAKED_INSTANCES). */
int pt_upload_dynamic_async(pt_ctx* ctx, const pt_emissive_triangle* lights, uint32_t n_lights,
    const pt_top_bvh_node* top_nodes, uint32_t n_top, uint32_t top_root);
int pt_frame_tick(pt_ctx* ctx);
/* New vertices and refitted boxes for an unchanged topology (refitBVH, src/bvh/refit_bvh.cpp:6-34; MeshSequence::buildBvh,
 * src/model/mesh_sequence.cpp:81-97): the caller's whole vertex and sub-BVH arrays after the refit.  Takes effect with the next
 * pt_upload_dynamic(_async) + pt_frame_tick. */
int pt_update_geometry(pt_ctx* ctx, const pt_vertex* verts, uint32_t n_verts, const pt_sub_bvh_node* nodes, uint32_t n_nodes);
/* The same refit WITHOUT the caller's nodes (round 5): `n_verts` vertex records replace [first_vertex, first_vertex + n_verts) of the array
 * pt_upload_static took -- the vertices of the mesh that moved -- and the device recomputes every box of its trees bottom-up from the
 * triangles (what refitBVH does on the host in the reference, src/bvh/refit_bvh.cpp:6-34) and re-makes the triangle records, on the copy
 * stream, nothing synchronised.  The host's share of a tick is one copy of the moved vertices.  The boxes are the bits pt_update_geometry
 * makes from host-refitted nodes.  PT_ERR_UNSUPPORTED when two roots of the sub-BVH array share a subtree (use pt_update_geometry).
 * Takes effect with the next pt_upload_dynamic(_async) + pt_frame_tick. */
int pt_refit_vertices(pt_ctx* ctx, uint32_t first_vertex, const pt_vertex* verts, uint32_t n_verts);
/* kind 0: material textures (CLTextureArray 1024x1024, CL_BGRA / CL_UNORM_INT8 in the reference, src/raytracer.cpp:284,
 * src/opencl/texture.cpp:112-131,148), kind 1: skydome (CL_RGBA / CL_FLOAT, src/raytracer.cpp:153-160, texture.cpp:96-110).
 * format: the two image formats the reference creates (texture.cpp:133-164) --
 *   PT_TEX_RGBA32F     data = layers*h*w*4 floats, r g b a, already linear / brightness-scaled (what read_imagef returns)
 *   PT_TEX_BGRA8_UNORM data = layers*h*w*4 bytes, b g r a (the FreeImage 32-bit bitmap the reference uploads); a fetch
 *                      returns byte / 255 like read_imagef on CL_UNORM_INT8; a quarter of the bytes per fetch.
 * Rows bottom-up as FreeImage stores them (the reference uploads FreeImage_GetBits as is).  Sampling reproduces
 * CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_REPEAT | CLK_FILTER_LINEAR. */
typedef enum { PT_TEX_RGBA32F = 0, PT_TEX_BGRA8_UNORM = 1 } pt_texture_format;
int pt_upload_texture_array(pt_ctx* ctx, int kind, uint32_t width, uint32_t height, uint32_t layers, int format, const void* data);

/* ---- per-frame state -- replaces the KernelData upload, src/raytracer.cpp:294-317 */
int pt_set_camera(pt_ctx* ctx, const pt_camera* cam); /* does NOT clear; caller decides (src/raytracer.cpp:99-105) */
/* Restrict this context to a set of image rectangles (multi-GPU tile sharding).  n = 0: whole image. */
int pt_set_tiles(pt_ctx* ctx, const pt_rect* rects, uint32_t n);
/* Use caller-owned device memory (width*height float4) as the HDR accumulator, e.g. a torch tensor
 * that is then reduced with torch.distributed / RCCL.  NULL = library-owned. */
int pt_set_accum_buffer(pt_ctx* ctx, void* device_float4);
int pt_clear(pt_ctx* ctx); /* clearAccumulationBuffer, src/raytracer.cpp:452-462; also resets spp */

/* ---- render -- replaces RayTracer::traceRays (src/raytracer.cpp:289-430): `spp` samples per owned pixel,
 * asynchronous on the context's stream.  The accumulator is bit-reproducible for the same sequence of pt_render calls (whatever
 * the schedule inside, the tile partition or the GPU count); another grouping of the same samples -- pt_render(N) against N x
 * pt_render(1) -- sums in another order and agrees to round-off only. */
int pt_render(pt_ctx* ctx, uint32_t spp);
int pt_synchronize(pt_ctx* ctx);
/* accumulate kernel, assets/cl/accumulate.cl:6-34: mean -> exposure -> Reinhard -> sRGB; width*height*4 floats out (host) */
int pt_resolve(pt_ctx* ctx, float* rgba_out);
/* same kernel, output left in DEVICE memory (width*height float4) -- the analogue of the GL texture the reference's
 * accumulate kernel writes (src/raytracer.cpp:432-450); NULL: a buffer owned by the context.  Asynchronous. */
int pt_resolve_device(pt_ctx* ctx, void* device_rgba);
void* pt_resolve_device_ptr(pt_ctx* ctx); /* the context-owned image pt_resolve_device(ctx, NULL) writes; NULL before the first such call */
int pt_read_accum(pt_ctx* ctx, float* out_float4); /* width*height*4 floats: HDR sums (host) */
int pt_write_accum(pt_ctx* ctx, const float* in_float4, uint32_t spp); /* restore a checkpoint */
void* pt_accum_device_ptr(pt_ctx* ctx);
uint32_t pt_samples_per_pixel(const pt_ctx* ctx); /* RayTracer::getSamplesPerPixel */
int pt_stats_get(pt_ctx* ctx, pt_stats* out); /* synchronises */
int pt_stats_reset(pt_ctx* ctx);
int pt_profile_kernels(pt_ctx* ctx, int enable); /* per-kernel hipEvent timing inside pt_render */
/* Sum the accumulator over ranks onto `root` with RCCL (ncclReduce, float sum). comm: ncclComm_t. */
int pt_reduce_accum(pt_ctx* ctx, void* nccl_comm, int root);

/* ---- kernel-granular entry points (parity tests and micro-benchmarks) ------------------
 * Host SoA arrays in, host SoA arrays out; the scene must have been uploaded. */
typedef struct {
    const float *ox, *oy, *oz, *dx, *dy, *dz; /* n each */
    const float* tmax; /* any-hit: ray length; closest-hit: ignored (INFINITY) */
} pt_rays_soa;
typedef struct {
    float *t, *u, *v; /* closest hit: t (INFINITY on miss), barycentrics */
    int32_t* prim; /* index into the (reordered) triangle array, -1 on miss; any-hit: 1/0 occluded */
    int32_t* inst; /* top-level LEAF node index of the hit instance, -1 on miss */
} pt_hits_soa;
/* traceRay (assets/cl/scene.cl:61-271) over n rays; any_hit selects intersectShadows semantics.
 * repeat > 1 re-runs the launch for timing; ms_out (optional) receives the mean device ms per launch. */
int pt_intersect(pt_ctx* ctx, const pt_rays_soa* rays, uint32_t n, int any_hit, pt_hits_soa* hits, uint32_t repeat, float* ms_out);
/* generatePrimaryRays (kernel.cl:24-84) for sample index `sample`: first n owned pixels, SoA out. */
int pt_gen_rays(pt_ctx* ctx, uint32_t sample, uint32_t n, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz, uint32_t* pixel);
/* generatePrimaryRays + the first intersectWalk of one batch (src/raytracer.cpp:323-357) exactly as pt_render issues them for `batch`
 * samples per owned pixel starting at sample index `sample` -- camera rays generated inside the traversal kernel and walked as bundles
 * where pt_render does that.  n = owned pixels * batch (<= 64 M: everything is read back); rays (optional) and hit records come back in
 * queue order.  Fixed schedule only. */
int pt_primary_pass(pt_ctx* ctx, uint32_t sample, uint32_t batch, uint32_t n, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz,
    uint32_t* pixel, pt_hits_soa* hits);

/* One shade invocation per entry (kernel.cl:190-301, neeIsShading shading.cl:356-623), PT_RNG_COUNTER keying. */
typedef struct {
    uint32_t n;
    /* in: the ray that was traced, its path state and its hit record */
    const float *ox, *oy, *oz, *dx, *dy, *dz;
    const float *thr_r, *thr_g, *thr_b;
    const uint32_t* pixel;
    const uint32_t* flags; /* bit1 = LASTSPECULAR */
    const uint32_t* bounce;
    const float *t, *u, *v;
    const int32_t *prim, *inst;
    uint32_t sample;
    /* out */
    float *radiance; /* 3n: deposit made by shade itself (emissive / sky) */
    uint32_t* out_alive; /* continuation ray spawned */
    float *nox, *noy, *noz, *ndx, *ndy, *ndz, *nthr_r, *nthr_g, *nthr_b;
    uint32_t* nflags;
    uint32_t* shadow_alive;
    float *sox, *soy, *soz, *sdx, *sdy, *sdz, *slen, *sc_r, *sc_g, *sc_b;
} pt_shade_batch_io;
int pt_shade_batch(pt_ctx* ctx, pt_shade_batch_io* io);

/* test hook, no device needed: the host side of the routine that quantises up to four child boxes (lo / hi: 4 x 3 floats; empty[k] != 0:
 * unused slot) into the 64-byte 4-wide node the traversal kernels read (out64).  Every child box must lie inside its quantised box. */
/* Diagnostic (bench.py, roofline.peak_measured_copy): GB/s, read + written, of a float4 grid-stride device copy of `bytes` bytes -- what the HBM of the box at hand delivers
 * to the access shape MI355X_MICROARCH.md measures 6.29 TB/s with (the reference has no counterpart). */
int pt_debug_copy_bandwidth(pt_ctx* ctx, size_t bytes, uint32_t repeat, float* gbps_out);
int pt_debug_quantise_node(const float* lo12, const float* hi12, const uint32_t* refs4, const uint8_t* empty4, uint32_t empty_ref, void* out64);
const char* pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
