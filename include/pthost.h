/*
 * pthost.h - C host side of the MI355X radiance path (libpthost.so, plain C, built with gcc).
 *
 * This is the part of the reference that stays on the host and stays in C (BASELINE.json
 * north_star): the text-scene reader, OBJ ingestion, camera basis, the rand()-seeded per-pixel
 * RNG state and the PPM snapshot.  Every entry point cites the reference code it replaces.
 * Nothing here touches the GPU; the device boundary is include/ptgpu.h.
 *
 * Error convention: functions return 0 on success and a negative value on failure, with a
 * message available from pthost_last_error().  (The reference prints to stderr and calls
 * exit(-1), e.g. displayfunc.c:107-110; the CLI in path-tracer_amd/host/ptcli.c reproduces
 * that print-and-exit behaviour on top of these status codes.)
 */
#ifndef PTHOST_H
#define PTHOST_H

#include "pt_records.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A loaded scene: what readScene() leaves in the reference's globals
 * (camera, objects, objectCount, lightCount, materials, scene; displayfunc.c:23-29). */
typedef struct pthost_scene {
    pt_scene     scene;         /* gamma, sky colours, material_count (3 defaults + user) */
    pt_camera    camera;        /* orig/target/width/height set, pitch = yaw = 0; basis NOT yet computed */
    pt_object   *objects;       /* object_count records, MODEL placeholders included */
    uint32_t     object_count;  /* after model expansion */
    uint32_t     light_count;   /* emissive spheres/triangles counted with the reference's viszero(x,x,z) */
    pt_material *materials;     /* scene.material_count records; [0..2] are the built-in types */
    char         title[100];    /* basename of the file without extension (displayfunc.c:112-113) */
} pthost_scene;

/* Replaces readScene(), reference displayfunc.c:103-406.
 *  path      : scene text file.
 *  model_dir : directory that relative `model` paths are resolved against; NULL = the current
 *              working directory, which is what the reference does (displayfunc.c:309).
 *  width,height : image size copied into camera.width/height (displayfunc.c:132-133).
 * Fields the reference leaves as uninitialised heap (default-material colours, MODEL placeholder
 * geometry) are zero here. */
int pthost_load_scene(const char *path, const char *model_dir, int width, int height,
                      pthost_scene *out);
void pthost_free_scene(pthost_scene *s);

/* Replaces updateCameraBasis(), reference camera.c:8-32 (double-promoted sin/cos/sqrt kept). */
void pthost_update_camera_basis(pt_camera *camera);

/* OBJ ingestion as the reference sees it through tinyobj_parse_obj(flags = 0) on the raw,
 * non-NUL-terminated file buffer (displayfunc.c:48-59, 309-316; tinyobj_loader_c.h:1149-1459):
 * vertices (3 floats each), the flat list of per-corner vertex indices (already zero-based),
 * and the number of `f` lines kept.  The last line of a file that ends in a single '\n' is
 * dropped, exactly as the reference drops it.  Buffers are malloc()ed; free with free(). */
int pthost_obj_load(const char *path, float **vertices, uint32_t *num_vertices,
                    int32_t **corner_vertex_index, uint32_t *num_corners, uint32_t *num_faces);
/* Same on an in-memory buffer (used by the tests for edge cases). */
int pthost_obj_parse(const char *buf, size_t len, float **vertices, uint32_t *num_vertices,
                     int32_t **corner_vertex_index, uint32_t *num_corners, uint32_t *num_faces);

/* Per-pixel RNG seeds, replaces the loop in allocateOutputBuffers(), reference
 * main-gpu.c:91-96 / main-cpu.c:44-49: seeds[i] = rand(); if (seeds[i] < 2) seeds[i] = 2;
 * The generator is a re-implementation of glibc's default-state rand() (TYPE_3 additive
 * feedback, seed 1) so that a render does not depend on who else called rand(); `skip`
 * outputs are discarded first.  Rank g of a multi-GPU render uses skip = g * count
 * (SURVEY.md 8-d "Seeds"). */
void pthost_seed_block(uint32_t *seeds, size_t count, uint64_t skip);

/* Row-flip a seed (2 words/pixel) or colour (3 floats/pixel) image in place: the reference's
 * CPU loop reads seeds at (H-1-y)*W+x while its kernel reads y*W+x. */
void pthost_flip_rows_u32x2(uint32_t *seeds, int width, int height);

/* Replaces the 'p' key handler, reference displayfunc.c:498-519: ASCII P3, rows bottom-up,
 * R = low byte of each pixel word. */
int pthost_write_ppm(const char *path, const uint32_t *pixels, int width, int height);

/* "<title>-<passes>-<elapsed>.ppm", the reference's snapshot name (displayfunc.c:500-503,
 * time-utils.c:17-32). */
void pthost_snapshot_name(char *out, size_t out_len, const char *title, int passes,
                          double elapsed_seconds);

/* Caption statistics, reference main-gpu.c:846-857. */
void pthost_captions(char *line1, char *line2, size_t len, double elapsed_total,
                     double elapsed_frame, int current_sample, int samples_this_frame,
                     int width, int height);

/* Adaptive passes-per-frame schedule of updateRendering(), reference main-gpu.c:774-794:
 * returns the time budget in seconds for the frame that starts at `current_sample`
 * (0 means "exactly one pass"). */
float pthost_frame_time_budget(int current_sample);

/* Checkpoint / resume (no reference counterpart - the reference's resumable state
 * is exactly these three things, main-gpu.c:23-25, 40): the running-mean colour buffer
 * [(H-1-y)*W+x], the per-pixel RNG state [2*(y*W+x)] and the sample counter.  The file is
 * "PTCKPT01", int32 width, height, current_sample, then 12*W*H bytes of colours and 8*W*H bytes of
 * seeds, little endian.  Resume = ptgpu_resize(w, h, seeds) + ptgpu_write_colors(colors) and
 * continue at current_sample: the continued render is bit-identical to an uninterrupted one.
 * pthost_checkpoint_read malloc()s *colors and *seeds. */
int pthost_checkpoint_write(const char *path, int width, int height, int current_sample,
                            const float *colors, const uint32_t *seeds);
int pthost_checkpoint_read(const char *path, int *width, int *height, int *current_sample,
                           float **colors, uint32_t **seeds);

const char *pthost_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTHOST_H */
