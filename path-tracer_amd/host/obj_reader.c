/*
 * obj_reader.c - Wavefront OBJ ingestion with the exact observable behaviour of the reference's
 * use of the vendored tinyobj_loader_c (MIT, Syoyo Fujita) at displayfunc.c:309-316:
 *   tinyobj_parse_obj(&attrib, ..., data, data_len, flags = 0)  on the raw file bytes.
 *
 * Only what the scene reader consumes is produced (displayfunc.c:326-348): vertex positions,
 * the flat per-corner v_idx list, and the `f`-line count.  Normals/texcoords are counted (they
 * take part in nothing we need) and materials/groups/shapes are ignored.
 *
 * Behaviours kept on purpose (tinyobj_loader_c.h line numbers):
 *  - the byte at len-1 is never examined when splitting lines (:1180-1203, "Assume last char is
 *    '\0'"), so the last line of a file that ends in ONE '\n' is dropped; a file ending "\n\n"
 *    keeps all its lines (cylinder 123/124 faces, teapot 239/240, ...);
 *  - numbers are read by the loader's own decimal routine (:267-393), not strtod: the fraction is
 *    summed digit by digit with repeated multiplication by 0.1, and a decimal exponent e is applied
 *    as 5^e * (1 << e); for negative e the reference's build multiplies by 1, i.e. it IGNORES
 *    negative exponents (bunny.obj's "-3.41e-003" loads as -3.41: the reference cannot load that
 *    file sensibly, and neither do we - same bytes);
 *  - `f` corners: "i", "i/j", "i//k", "i/j/k"; positive indices are 1-based, 0 stays 0, negative
 *    ones are relative to the vertices seen so far (:176-180).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pthost.h"
#include "pthost_internal.h"

#define OBJ_MAX_LINE 4095          /* parseLine's linebuf, tinyobj_loader_c.h:960-962 */
#define OBJ_MAX_CORNERS_PER_FACE 16 /* TINYOBJ_MAX_FACES_PER_F_LINE */

static int is_blank(char c) { return c == ' ' || c == '\t'; }
static int is_digit(char c) { return (unsigned)(c - '0') < 10u; }

/* tinyobj's tryParseDouble (tinyobj_loader_c.h:267-393): returns 1 and sets *out on success. */
static int obj_parse_decimal(const char *s, const char *end, double *out) {
    double mant = 0.0;
    int expo = 0, ndig = 0;
    char sgn = '+', esgn = '+';
    const char *p = s;

    if (s >= end) return 0;
    if (*p == '+' || *p == '-') {
        sgn = *p++;
    } else if (!is_digit(*p)) {
        return 0;
    }
    while (p != end && is_digit(*p)) {
        mant *= 10;
        mant += (int)(*p - '0');
        ++p;
        ++ndig;
    }
    if (ndig == 0) return 0;
    if (p == end) goto assemble;

    if (*p == '.') {
        int pos = 1;
        ++p;
        while (p != end && is_digit(*p)) {
            double w = 1.0;
            for (int k = 0; k < pos; ++k) w *= 0.1;
            mant += (int)(*p - '0') * w;
            ++pos;
            ++p;
        }
    } else if (*p != 'e' && *p != 'E') {
        goto assemble;
    }
    if (p == end) goto assemble;

    if (*p == 'e' || *p == 'E') {
        ++p;
        if (p != end && (*p == '+' || *p == '-')) {
            esgn = *p++;
        } else if (!is_digit(*p)) { /* the loader peeks *p even at end; linebuf is NUL-terminated */
            return 0;
        }
        ndig = 0;
        while (p != end && is_digit(*p)) {
            expo = expo * 10 + (int)(*p - '0');
            ++p;
            ++ndig;
        }
        if (esgn != '+') expo = -expo;
        if (ndig == 0) return 0;
    }

assemble: {
    double five = 1.0;
    for (int k = 0; k < expo; ++k) five = five * 5.0;
    /* (1 << expo) is undefined in C for expo < 0 or expo > 31.  Pinned to what the reference's
     * gcc 11.4 -O3 x86-64 build does (probed through oracle/_ref, tests/test_host_reader.py):
     * a negative exponent multiplies by 1, a large one takes the count modulo 32. */
    int two = expo < 0 ? 1 : (int)(1u << ((unsigned)expo & 31u));
    *out = (sgn == '+' ? 1 : -1) * (mant * five) * (double)two;
}
    return 1;
}

static int span_until_space(const char *t) {
    const char *p = t;
    while (*p != '\0' && *p != ' ' && *p != '\t' && *p != '\r') ++p;
    return (int)(p - t);
}

static float obj_next_float(const char **tok) {
    double v = 0.0;
    while (is_blank(**tok)) ++*tok;
    const char *end = *tok + span_until_space(*tok);
    obj_parse_decimal(*tok, end, &v);
    *tok = end;
    return (float)v;
}

static int obj_atoi(const char *c) {
    int v = 0, s = 1;
    if (*c == '+' || *c == '-') {
        if (*c == '-') s = -1;
        ++c;
    }
    while (*c >= '0' && *c <= '9') {
        v = v * 10 + (int)(*c - '0');
        ++c;
    }
    return v * s;
}

static void skip_index_field(const char **tok) {
    while (**tok != '\0' && **tok != '/' && **tok != ' ' && **tok != '\t' && **tok != '\r') ++*tok;
}

/* one "v", "v/t", "v//n" or "v/t/n" corner; only the position index is returned */
static int obj_corner(const char **tok) {
    int v = obj_atoi(*tok);
    skip_index_field(tok);
    if (**tok != '/') return v;
    ++*tok;
    if (**tok == '/') { /* v//n */
        ++*tok;
        skip_index_field(tok);
        return v;
    }
    skip_index_field(tok); /* t */
    if (**tok != '/') return v;
    ++*tok;
    skip_index_field(tok); /* n */
    return v;
}

static int line_ends_at(const char *b, size_t i, size_t last) {
    if (b[i] == '\0' || b[i] == '\n') return 1;
    if (b[i] == '\r' && i + 1 < last && b[i + 1] != '\n') return 1;
    return 0;
}

typedef struct {
    float   *v;
    uint32_t nv, capv;
    int32_t *idx;
    uint32_t nidx, capidx;
    uint32_t nfaces;
} obj_acc;

static int acc_push_vertex(obj_acc *a, float x, float y, float z) {
    if (a->nv == a->capv) {
        uint32_t nc = a->capv ? a->capv * 2 : 256;
        float *nvp = (float *)realloc(a->v, sizeof(float) * 3 * nc);
        if (!nvp) return -1;
        a->v = nvp;
        a->capv = nc;
    }
    a->v[3 * a->nv + 0] = x;
    a->v[3 * a->nv + 1] = y;
    a->v[3 * a->nv + 2] = z;
    a->nv++;
    return 0;
}

static int acc_push_index(obj_acc *a, int32_t i) {
    if (a->nidx == a->capidx) {
        uint32_t nc = a->capidx ? a->capidx * 2 : 768;
        int32_t *np = (int32_t *)realloc(a->idx, sizeof(int32_t) * nc);
        if (!np) return -1;
        a->idx = np;
        a->capidx = nc;
    }
    a->idx[a->nidx++] = i;
    return 0;
}

int pthost_obj_parse(const char *buf, size_t len, float **vertices, uint32_t *num_vertices,
                     int32_t **corner_vertex_index, uint32_t *num_corners, uint32_t *num_faces) {
    obj_acc a;
    memset(&a, 0, sizeof a);
    if (!buf || len < 1) {
        pthost_set_error("obj: empty buffer");
        return -1;
    }
    const size_t last = len - 1; /* never examined, see header comment */
    size_t start = 0, nlines = 0;
    char line[OBJ_MAX_LINE + 1];

    for (size_t i = 0; i < last; ++i) {
        if (!line_ends_at(buf, i, last)) continue;
        size_t n = i - start;
        const char *src = buf + start;
        start = i + 1;
        ++nlines;
        if (n >= OBJ_MAX_LINE) {
            pthost_set_error("obj: line longer than %d bytes", OBJ_MAX_LINE);
            goto fail;
        }
        memcpy(line, src, n);
        line[n] = '\0';

        const char *t = line;
        while (is_blank(*t)) ++t;
        if (*t == '\0' || *t == '#') continue;

        if (t[0] == 'v' && is_blank(t[1])) {
            t += 2;
            float x = obj_next_float(&t);
            float y = obj_next_float(&t);
            float z = obj_next_float(&t);
            if (acc_push_vertex(&a, x, y, z)) goto oom;
        } else if (t[0] == 'f' && is_blank(t[1])) {
            int32_t corner[OBJ_MAX_CORNERS_PER_FACE];
            int nc = 0;
            t += 2;
            while (is_blank(*t)) ++t;
            while (*t != '\r' && *t != '\n' && *t != '\0') {
                if (nc >= OBJ_MAX_CORNERS_PER_FACE - 1) { /* the loader asserts num_f < 16 */
                    pthost_set_error("obj: face with %d or more corners", OBJ_MAX_CORNERS_PER_FACE);
                    goto fail;
                }
                corner[nc++] = obj_corner(&t);
                while (is_blank(*t) || *t == '\r') ++t;
            }
            for (int k = 0; k < nc; ++k) {
                int32_t raw = corner[k];
                int32_t zero_based = raw > 0 ? raw - 1 : (raw == 0 ? 0 : (int32_t)a.nv + raw);
                if (acc_push_index(&a, zero_based)) goto oom;
            }
            a.nfaces++;
        }
        /* vn / vt / usemtl / mtllib / g / o / s: nothing the scene reader consumes */
    }
    if (nlines == 0) {
        pthost_set_error("obj: no complete line in buffer");
        goto fail;
    }
    *vertices = a.v;
    *num_vertices = a.nv;
    *corner_vertex_index = a.idx;
    *num_corners = a.nidx;
    *num_faces = a.nfaces;
    return 0;
oom:
    pthost_set_error("obj: out of memory");
fail:
    free(a.v);
    free(a.idx);
    return -1;
}

int pthost_obj_load(const char *path, float **vertices, uint32_t *num_vertices,
                    int32_t **corner_vertex_index, uint32_t *num_corners, uint32_t *num_faces) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        pthost_set_error("obj: cannot open '%s'", path);
        return -1;
    }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    rewind(f);
    if (sz < 1) {
        fclose(f);
        pthost_set_error("obj: '%s' is empty", path);
        return -1;
    }
    char *buf = (char *)malloc((size_t)sz);
    if (!buf) {
        fclose(f);
        pthost_set_error("obj: out of memory");
        return -1;
    }
    size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    int rc = pthost_obj_parse(buf, got, vertices, num_vertices, corner_vertex_index, num_corners,
                              num_faces);
    free(buf);
    return rc;
}
