/*
 * ref_tinyobj.c - TEST INFRASTRUCTURE.  Compiles the vendored tinyobj_loader_c.h where it lies in
 * /root/reference and calls it the way the reference does (displayfunc.c:48-59, 302-326): the raw
 * file bytes, not NUL-terminated, flags = 0.  Used only to check libpthost's OBJ reader.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TINYOBJ_LOADER_C_IMPLEMENTATION
#include "tinyobj_loader_c.h"

/* returns 0 on success; caller frees *vertices and *vidx with free() */
int ref_tinyobj_load(const char *path, float **vertices, unsigned *num_vertices, int **vidx,
                     unsigned *num_corners, unsigned *num_faces) {
    FILE *file = fopen(path, "rb");
    if (!file) return -1;
    fseek(file, 0, SEEK_END);
    size_t len = (size_t)ftell(file);
    rewind(file);
    char *data = malloc(len);
    if (fread(data, 1, len, file) != len) {
        fclose(file);
        free(data);
        return -2;
    }
    fclose(file);

    tinyobj_attrib_t attrib;
    tinyobj_shape_t *shapes = NULL;
    size_t num_shapes;
    tinyobj_material_t *materials = NULL;
    size_t num_materials;
    int ret = tinyobj_parse_obj(&attrib, &shapes, &num_shapes, &materials, &num_materials, data, len, 0);
    free(data);
    if (ret != TINYOBJ_SUCCESS) return -3;

    *num_vertices = attrib.num_vertices;
    *num_corners = attrib.num_faces;
    *num_faces = attrib.num_face_num_verts;
    *vertices = malloc(sizeof(float) * 3 * (attrib.num_vertices ? attrib.num_vertices : 1));
    memcpy(*vertices, attrib.vertices, sizeof(float) * 3 * attrib.num_vertices);
    *vidx = malloc(sizeof(int) * (attrib.num_faces ? attrib.num_faces : 1));
    for (unsigned i = 0; i < attrib.num_faces; i++) (*vidx)[i] = attrib.faces[i].v_idx;
    tinyobj_attrib_free(&attrib);
    tinyobj_shapes_free(shapes, num_shapes);
    tinyobj_materials_free(materials, num_materials);
    return 0;
}
