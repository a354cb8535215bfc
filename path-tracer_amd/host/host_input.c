/*
 * host_input.c - the reference's interactive input, headless: what its GLUT callbacks do to the camera, the
 * sample counter and the image buffers (displayfunc.c:408-447 updateCamera, :450-458 update, :482-493 reshape,
 * :496-592 keyboard, :600-622 keyboardUp, :628-659 motion / mouse, :661-671 special / specialUp) and
 * reInitViewpointDependentBuffers (main-gpu.c:885-913), restated as plain functions over a state record.
 * No window: the caller (ptcli --script, a test) feeds the events and acts on the flags that come back.
 *
 * Arithmetic is the reference host build's, expression by expression: float variables, C's promotions
 * (sqrt / floor on float arguments and the 0.01 literals are double), no contraction (Makefile: -ffp-contract=off).
 * Parity: displayfunc.c cannot be built here (it needs GL/freeglut.h), so these functions are pinned by reading only
 * - "parity unpinned" - except for the camera basis every event ends in, which is camera.c's and is checked
 * against the reference's own updateCameraBasis (oracle/_ref) after each event (tests/test_host_input.py).
 */
#include <math.h>
#include <string.h>

#include "pthost.h"

#define PI_OVER_TWO_F 1.57079632679489661923f /* geom.h:8 */
#define TWO_PI_F 6.28318530717958647693f      /* geom.h:9 */
#define GLUT_LEFT_BUTTON_VALUE 0              /* freeglut_std.h; mouseButton starts at 0 (displayfunc.c:41) */

/* vnorm, vec.h:20: float l = 1.f / sqrt(vdot(v, v)) - the square root and the division in double */
static void vnorm_promoted(pt_vec3 *v) {
    const float l = (float)(1.f / sqrt((double)(v->x * v->x + v->y * v->y + v->z * v->z)));
    v->x = l * v->x;
    v->y = l * v->y;
    v->z = l * v->z;
}

/* reInitViewpointDependentBuffers(0), main-gpu.c:885-913: basis, (camera upload), currentSample = 0 */
static int viewpoint_changed(pthost_input *in) {
    pthost_update_camera_basis(&in->camera);
    return PTHOST_EV_CAMERA;
}

void pthost_input_init(pthost_input *in, const pt_camera *loaded_camera, int width, int height) {
    memset(in, 0, sizeof *in);
    in->camera = *loaded_camera;
    in->original_camera = *loaded_camera; /* displayfunc.c:135, before any basis is computed */
    in->width = width;
    in->height = height;
    /* setUpOpenCL -> allocateOutputBuffers has drawn the first 2*W*H rand() outputs (main-gpu.c:91-96) */
    in->rand_consumed = 2ull * (unsigned long long)width * (unsigned long long)height;
    pthost_update_camera_basis(&in->camera); /* main-gpu.c:950 */
}

/* updateCamera, displayfunc.c:408-444 */
static int update_camera(pthost_input *in, int delta) {
    int moved = 0;
    const float speed = delta * 0.0001f * (in->shift_pressed ? 10 : 1);
    pt_vec3 movement = {0.f, 0.f, 0.f};
    const unsigned char *k = in->key_states;
    if (k['w'] || k['W'] || k['s'] || k['S']) {
        const float front_or_back = (k['w'] || k['W']) ? 1 : -1;
        movement.x = front_or_back * in->camera.dir.x;
        movement.y = front_or_back * in->camera.dir.y;
        movement.z = front_or_back * in->camera.dir.z;
        moved = 1;
    }
    if (k['a'] || k['A'] || k['d'] || k['D']) {
        const float left_or_right = (k['d'] || k['D']) ? 1 : -1;
        pt_vec3 direction = in->camera.x;
        vnorm_promoted(&direction);
        direction.x = left_or_right * direction.x;
        direction.y = left_or_right * direction.y;
        direction.z = left_or_right * direction.z;
        movement.x = direction.x + movement.x;
        movement.y = direction.y + movement.y;
        movement.z = direction.z + movement.z;
        moved = 1;
    }
    if (!moved) return PTHOST_EV_NONE;
    vnorm_promoted(&movement);
    movement.x = speed * movement.x;
    movement.y = speed * movement.y;
    movement.z = speed * movement.z;
    in->camera.orig.x = in->camera.orig.x + movement.x;
    in->camera.orig.y = in->camera.orig.y + movement.y;
    in->camera.orig.z = in->camera.orig.z + movement.z;
    in->camera.target.x = in->camera.target.x + movement.x;
    in->camera.target.y = in->camera.target.y + movement.y;
    in->camera.target.z = in->camera.target.z + movement.z;
    return viewpoint_changed(in);
}

/* update(), displayfunc.c:450-458, up to the call of updateRendering().  `delta = timeNow - (timeBefore || 0)`
 * is a LOGICAL or upstream: the subtrahend is 0 or 1, so the step grows with the time since GLUT started. */
int pthost_input_idle(pthost_input *in, int time_now_ms) {
    const int delta = time_now_ms - (in->time_before || 0);
    const int ev = update_camera(in, delta);
    in->time_before = time_now_ms;
    return ev;
}

static void set_key_states(pthost_input *in, unsigned char lower, unsigned char upper, unsigned char value) {
    in->key_states[lower] = value;
    in->key_states[upper] = value;
}

/* keyboard(), displayfunc.c:496-592 */
int pthost_input_keyboard(pthost_input *in, unsigned char key) {
    int ev = PTHOST_EV_NONE;
    switch (key) {
    case 'P': case 'p': ev = PTHOST_EV_SNAPSHOT; break;
    case ' ': /* back to the camera the scene file gave */
        in->camera = in->original_camera;
        ev = viewpoint_changed(in);
        break;
    case 27: ev = PTHOST_EV_QUIT; break;
    case 'w': case 'W': set_key_states(in, 'w', 'W', 1); break;
    case 's': case 'S': set_key_states(in, 's', 'S', 1); break;
    case 'a': case 'A': set_key_states(in, 'a', 'A', 1); break;
    case 'd': case 'D': set_key_states(in, 'd', 'D', 1); break;
    default: break;
    }
    in->key_states[key] = 1;
    return ev;
}

/* keyboardUp(), displayfunc.c:600-622: only the four movement keys are ever released */
int pthost_input_keyboard_up(pthost_input *in, unsigned char key) {
    switch (key) {
    case 'w': case 'W': set_key_states(in, 'w', 'W', 0); break;
    case 's': case 'S': set_key_states(in, 's', 'S', 0); break;
    case 'a': case 'A': set_key_states(in, 'a', 'A', 0); break;
    case 'd': case 'D': set_key_states(in, 'd', 'D', 0); break;
    default: break;
    }
    return PTHOST_EV_NONE;
}

/* special() / specialUp(), displayfunc.c:661-671, for the two shift keys */
int pthost_input_shift(pthost_input *in, int pressed) {
    in->shift_pressed = pressed ? 1 : 0;
    return PTHOST_EV_NONE;
}

/* motion(), displayfunc.c:628-650 */
int pthost_input_motion(pthost_input *in, int x, int y) {
    const int delta_x = in->mouse_x - x;
    const int delta_y = in->mouse_y - y;
    if (delta_x == 0 && delta_y == 0) return PTHOST_EV_NONE;
    if (in->mouse_button == GLUT_LEFT_BUTTON_VALUE) {
        pt_camera *c = &in->camera;
        c->yaw += delta_x * 0.01;
        c->yaw = c->yaw - TWO_PI_F * floor(c->yaw / TWO_PI_F);
        c->pitch += -delta_y * 0.01;
        c->pitch = c->pitch < -PI_OVER_TWO_F ? -PI_OVER_TWO_F : (c->pitch > PI_OVER_TWO_F ? PI_OVER_TWO_F : c->pitch);
    }
    in->mouse_x = x;
    in->mouse_y = y;
    return viewpoint_changed(in); /* (whatever the button: displayfunc.c:646) */
}

/* mouse(), displayfunc.c:653-659: records button and position; the motion() call it ends in sees no movement */
int pthost_input_mouse(pthost_input *in, int button, int x, int y) {
    in->mouse_button = button;
    in->mouse_x = x;
    in->mouse_y = y;
    return pthost_input_motion(in, x, y);
}

/* reshape(), displayfunc.c:482-493 -> reInitViewpointDependentBuffers(1), main-gpu.c:887-890: the image buffers
 * are re-allocated at the new size with FRESH rand() seeds (pthost_input_next_seeds), the basis is recomputed -
 * from camera.width / camera.height, which reshape does NOT touch (they keep the size given on the command
 * line, displayfunc.c:132-133: the aspect ratio of the view stays the first window's). */
int pthost_input_reshape(pthost_input *in, int new_width, int new_height) {
    in->width = new_width;
    in->height = new_height;
    return viewpoint_changed(in) | PTHOST_EV_REALLOC;
}

/* allocateOutputBuffers' seed loop for the current window size, continuing the process's rand() stream
 * (main-gpu.c:91-96; the reference never calls srand) */
void pthost_input_next_seeds(pthost_input *in, uint32_t *seeds) {
    const size_t count = 2 * (size_t)in->width * (size_t)in->height;
    pthost_seed_block(seeds, count, in->rand_consumed);
    in->rand_consumed += count;
}
