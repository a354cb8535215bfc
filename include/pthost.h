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

/* ---- interactive input, headless -------------------------------------------------------
 * What the reference's GLUT callbacks do to the camera, the sample counter and the image buffers, as plain
 * functions over a state record: no window, the caller feeds the events (ptcli --script, the tests).  Every
 * function returns a mask of PTHOST_EV_*: what the caller has to do before the next frame.
 *   CAMERA    reInitViewpointDependentBuffers(0), main-gpu.c:885-913: `camera` (basis recomputed) is to be
 *             uploaded (ptgpu_set_camera) and the sample counter reset to 0
 *   REALLOC   reInitViewpointDependentBuffers(1), main-gpu.c:887-890: the image buffers are re-allocated at
 *             width x height with fresh rand() seeds (pthost_input_next_seeds + ptgpu_resize); comes with CAMERA
 *   SNAPSHOT  the 'p' key: write the PPM (pthost_snapshot_name + pthost_write_ppm)
 *   QUIT      ESC */
enum { PTHOST_EV_NONE = 0, PTHOST_EV_CAMERA = 1, PTHOST_EV_REALLOC = 2, PTHOST_EV_SNAPSHOT = 4, PTHOST_EV_QUIT = 8 };
typedef struct pthost_input {
    pt_camera camera;            /* the reference's global `camera`, basis up to date */
    pt_camera original_camera;   /* `originalCamera`, displayfunc.c:45, 135: what ' ' restores */
    unsigned char key_states[256]; /* displayfunc.c:42 */
    int shift_pressed, mouse_button, mouse_x, mouse_y, time_before; /* displayfunc.c:40-46 */
    int width, height;           /* window size (displayfunc.c:34-35); camera.width / height keep the first size */
    unsigned long long rand_consumed; /* rand() outputs drawn so far by seed (re)allocations */
} pthost_input;
/* State after main() has run readScene, updateCameraBasis and setUpOpenCL (main-gpu.c:948-956): `loaded_camera` as
 * pthost_load_scene left it, the first seed block (2*W*H rand() outputs) drawn. */
void pthost_input_init(pthost_input *in, const pt_camera *loaded_camera, int width, int height);
/* update() up to its updateRendering() call, displayfunc.c:450-453 + updateCamera :408-444: WASD held down move
 * `orig` and `target` along dir / x by delta * 1e-4 (x 10 with shift); time_now_ms = glutGet(GLUT_ELAPSED_TIME). */
int pthost_input_idle(pthost_input *in, int time_now_ms);
int pthost_input_keyboard(pthost_input *in, unsigned char key);     /* displayfunc.c:496-592 */
int pthost_input_keyboard_up(pthost_input *in, unsigned char key);  /* displayfunc.c:600-622 */
int pthost_input_shift(pthost_input *in, int pressed);              /* displayfunc.c:661-671 */
int pthost_input_mouse(pthost_input *in, int button, int x, int y); /* displayfunc.c:653-659 */
int pthost_input_motion(pthost_input *in, int x, int y);            /* displayfunc.c:628-650: yaw / pitch by 0.01 rad per pixel */
int pthost_input_reshape(pthost_input *in, int new_width, int new_height); /* displayfunc.c:482-493 */
/* The seeds of the (re)allocated buffers, 2 * width * height words: the next block of the process's rand() stream
 * (main-gpu.c:91-96).  GLUT fires reshape when the window is created, so an interactive run of the reference renders
 * with the SECOND block; a run that never reshapes (this library's headless default) with the first. */
void pthost_input_next_seeds(pthost_input *in, uint32_t *seeds);

const char *pthost_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTHOST_H */
