/*
 * Scene ingest: .rsc (JSON) scene descriptions and the Wavefront OBJ meshes
 * they name, turned into the scene arrays the renderer uploads.
 *
 * Behaviour follows the reference loader (src/loader.c:100-458) and the OBJ
 * parser it vendors (libs/tinyobj_loader_c.h, flag TRIANGULATE), including the
 * parts that decide which bits end up in the vertex array: the parser's own
 * decimal-to-double conversion (digit by digit, fraction digits weighted by a
 * repeated product of 0.1), the line splitting that ignores a last line without
 * a line ending, the index fix-up against the running element counts, and the
 * shape boundaries counted in face lines.  What the reference leaves undefined
 * is given a definition here and listed next to the code.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <float.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "loader.h"

static void fatal(const char *msg, const char *arg)
{
    if (arg) printf("Error: %s '%s'.\n", msg, arg);
    else printf("Error: %s.\n", msg);
    fflush(stdout);
    exit(1);
}

/* ------------------------------------------------------------------------- */
/* files                                                                     */
/* ------------------------------------------------------------------------- */
static char g_root[PATH_MAX];
static int g_root_set = 0;

void rt_set_resource_root(const char *dir)
{
    if (!dir) { g_root_set = 0; return; }
    snprintf(g_root, sizeof g_root, "%s", dir);
    g_root_set = 1;
}

static const char *resource_root(void)
{
    static char derived[PATH_MAX];
    if (g_root_set) return g_root;
    const char *env = getenv("RT_RES_DIR");
    if (env && env[0]) return env;
    char exe[PATH_MAX];
    ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (n <= 0) fatal("couldn't get bin path", NULL);
    exe[n] = '\0';
    char *slash = strrchr(exe, '/');
    if (slash) *slash = '\0';
    if (snprintf(derived, sizeof derived, "%s/res", exe) >= (int)sizeof derived) fatal("bin path too long", NULL);
    return derived;
}

char *load_file(const char *url, long *ret_length)
{
    char real_url[PATH_MAX];
    if (snprintf(real_url, sizeof real_url, "%s/%s", resource_root(), url) >= (int)sizeof real_url)
        fatal("path too long for", url);
    FILE *f = fopen(real_url, "rb");
    if (!f) fatal("Couldn't load file", real_url);
    fseek(f, 0, SEEK_END);
    long length = ftell(f) + 1;
    fseek(f, 0, SEEK_SET);
    char *buffer = (char *)malloc((size_t)length);
    if (!buffer) fatal("out of memory loading", real_url);
    size_t got = fread(buffer, 1, (size_t)length - 1, f);
    fclose(f);
    buffer[got] = '\0';
    *ret_length = (long)got + 1;
    return buffer;
}

/* ------------------------------------------------------------------------- */
/* OBJ                                                                       */
/* ------------------------------------------------------------------------- */
#define OBJ_IS_SPACE(c) ((c) == ' ' || (c) == '\t')
#define OBJ_IS_DIGIT(c) ((unsigned int)((c) - '0') < 10u)
#define OBJ_IS_NEW_LINE(c) ((c) == '\r' || (c) == '\n' || (c) == '\0')
#define OBJ_NO_INDEX INT_MIN          /* an absent vt / vn reference */
#define OBJ_MAX_CORNERS 1024

static void skip_blank(const char **t) { while (OBJ_IS_SPACE((*t)[0])) (*t)++; }
static void skip_blank_cr(const char **t) { while (OBJ_IS_SPACE((*t)[0]) || (*t)[0] == '\r') (*t)++; }

static int word_length(const char *t)
{
    const char *p = t;
    while (p[0] != '\0' && p[0] != ' ' && p[0] != '\t' && p[0] != '\r') p++;
    return (int)(p - t);
}

/* Decimal text -> double the way the vendored parser does it
   (libs/tinyobj_loader_c.h:266-392): no rounding-correct conversion, every
   digit is added in turn.  The sequence of floating-point operations decides
   the bits of every vertex, so this function follows tryParseDouble of
   tinyobj_loader_c operation for operation; that code is

     Copyright (c) 2016 Syoyo Fujita and many contributors, The MIT License (MIT).
     Permission is hereby granted, free of charge, to any person obtaining a copy
     of this software and associated documentation files (the "Software"), to deal
     in the Software without restriction, including without limitation the rights
     to use, copy, modify, merge, publish, distribute, sublicense, and/or sell
     copies of the Software, and to permit persons to whom the Software is
     furnished to do so, subject to the following conditions:
     The above copyright notice and this permission notice shall be included in
     all copies or substantial portions of the Software.
     THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR
     IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY,
     FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE
     AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER
     LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM,
     OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN
     THE SOFTWARE.
  Returns 0 when the text is not a number (the caller
   then uses 0.0).
   Defined here, undefined in the reference: a negative exponent or one above 30
   (the reference shifts 1 by the exponent); the value is mantissa * 10^exponent
   with the power built by repeated multiplication / division. */
static int parse_decimal(const char *s, const char *s_end, double *result)
{
    double mantissa = 0.0;
    int exponent = 0, read = 0, more;
    char sign = '+', exp_sign = '+';
    const char *c = s;

    if (s >= s_end) return 0;
    if (*c == '+' || *c == '-') { sign = *c; c++; }
    else if (!OBJ_IS_DIGIT(*c)) return 0;

    more = (c != s_end);
    while (more && OBJ_IS_DIGIT(*c)) {
        mantissa *= 10;
        mantissa += (int)(*c - '0');
        c++; read++;
        more = (c != s_end);
    }
    if (read == 0) return 0;
    if (!more) goto assemble;

    if (*c == '.') {
        c++;
        read = 1;
        more = (c != s_end);
        while (more && OBJ_IS_DIGIT(*c)) {
            double weight = 1.0;
            for (int f = 0; f < read; f++) weight *= 0.1;
            mantissa += (int)(*c - '0') * weight;
            read++; c++;
            more = (c != s_end);
        }
    } else if (*c != 'e' && *c != 'E') {
        goto assemble;
    }
    if (!more) goto assemble;

    if (*c == 'e' || *c == 'E') {
        c++;
        more = (c != s_end);
        if (more && (*c == '+' || *c == '-')) { exp_sign = *c; c++; }
        else if (!OBJ_IS_DIGIT(*c)) return 0;
        read = 0;
        more = (c != s_end);
        while (more && OBJ_IS_DIGIT(*c)) {
            exponent *= 10;
            exponent += (int)(*c - '0');
            c++; read++;
            more = (c != s_end);
        }
        if (exp_sign == '-') exponent = -exponent;
        if (read == 0) return 0;
    }

assemble:
    {
        double v;
        if (exponent >= 0 && exponent <= 30) {
            double a = 1.0;
            for (int i = 0; i < exponent; i++) a = a * 5.0;
            v = (mantissa * a) * (double)(1 << exponent);
        } else if (exponent > 30) {
            v = mantissa;
            for (int i = 0; i < exponent && v <= DBL_MAX; i++) v *= 10.0;
        } else {
            v = mantissa;
            for (int i = 0; i > exponent && v != 0.0; i--) v /= 10.0;
        }
        *result = (sign == '+' ? 1 : -1) * v;
    }
    return 1;
}

static float parse_float(const char **t)
{
    double v = 0.0;
    skip_blank(t);
    const char *end = (*t) + word_length(*t);
    parse_decimal(*t, end, &v);
    *t = end;
    return (float)v;
}

static int parse_int_prefix(const char *c)
{
    unsigned int value = 0;
    int negative = 0;
    if (*c == '+' || *c == '-') { negative = (*c == '-'); c++; }
    while (*c >= '0' && *c <= '9') { value = value * 10u + (unsigned int)(*c - '0'); c++; }
    return negative ? (int)(0u - value) : (int)value;
}

static void skip_index_text(const char **t)
{
    while ((*t)[0] != '\0' && (*t)[0] != '/' && (*t)[0] != ' ' && (*t)[0] != '\t' && (*t)[0] != '\r') (*t)++;
}

/* i, i/j, i/j/k, i//k  (libs/tinyobj_loader_c.h:182-228) */
static void parse_corner(const char **t, int *v, int *vt, int *vn)
{
    *v = *vt = *vn = OBJ_NO_INDEX;
    *v = parse_int_prefix(*t);
    skip_index_text(t);
    if ((*t)[0] != '/') return;
    (*t)++;
    if ((*t)[0] == '/') {
        (*t)++;
        *vn = parse_int_prefix(*t);
        skip_index_text(t);
        return;
    }
    *vt = parse_int_prefix(*t);
    skip_index_text(t);
    if ((*t)[0] != '/') return;
    (*t)++;
    *vn = parse_int_prefix(*t);
    skip_index_text(t);
}

/* one-based -> zero-based, negative = relative to the elements read so far
   (libs/tinyobj_loader_c.h:175-179); an absent reference goes through the same
   arithmetic, with wrap-around instead of signed overflow */
static int fix_index(int idx, size_t n)
{
    if (idx > 0) return idx - 1;
    if (idx == 0) return 0;
    return (int)((unsigned int)n + (unsigned int)idx);
}

typedef struct { void *p; size_t n, cap, elem; } grow;
static void *grow_push(grow *g, size_t count)
{
    if (g->n + count > g->cap) {
        size_t cap = g->cap ? g->cap * 2 : 1024;
        while (cap < g->n + count) cap *= 2;
        g->p = realloc(g->p, cap * g->elem);
        if (!g->p) fatal("out of memory while parsing a mesh", NULL);
        g->cap = cap;
    }
    void *at = (char *)g->p + g->n * g->elem;
    g->n += count;
    return at;
}

int rt_obj_parse(rt_obj *out, const char *buf, size_t len)
{
    memset(out, 0, sizeof *out);
    if (!buf || len < 1) return 1;

    grow V = {0, 0, 0, 3 * sizeof(float)}, N = {0, 0, 0, 3 * sizeof(float)}, T = {0, 0, 0, 2 * sizeof(float)};
    grow F = {0, 0, 0, 3 * sizeof(int)};
    grow S_off = {0, 0, 0, sizeof(unsigned int)}, S_len = {0, 0, 0, sizeof(unsigned int)};

    /* shape bookkeeping, in face lines (libs/tinyobj_loader_c.h:1367-1458) */
    unsigned int face_lines = 0, nontri = 0;
    unsigned int prev_shape_face_offset = 0, prev_face_offset = 0;
    size_t shape_idx = 0, num_lines = 0;

    char *line = NULL;
    size_t line_cap = 0;
    const size_t end_idx = len - 1;      /* the byte at len-1 is the terminating NUL and is not looked at */
    size_t prev_pos = 0;
    for (size_t i = 0; i < end_idx; i++) {
        const char c = buf[i];
        int ending = (c == '\0' || c == '\n');
        if (c == '\r' && (i + 1) < end_idx && buf[i + 1] != '\n') ending = 1;
        if (!ending) continue;
        const size_t llen = i - prev_pos;
        if (llen + 1 > line_cap) { line_cap = (llen + 1) * 2; line = (char *)realloc(line, line_cap); }
        memcpy(line, buf + prev_pos, llen);
        line[llen] = '\0';
        prev_pos = i + 1;
        num_lines++;

        const char *t = line;
        skip_blank(&t);
        if (t[0] == '\0' || t[0] == '#') continue;

        if (t[0] == 'v' && OBJ_IS_SPACE(t[1])) {
            t += 2;
            float *v = (float *)grow_push(&V, 1);
            v[0] = parse_float(&t); v[1] = parse_float(&t); v[2] = parse_float(&t);
        } else if (t[0] == 'v' && t[1] == 'n' && OBJ_IS_SPACE(t[2])) {
            t += 3;
            float *v = (float *)grow_push(&N, 1);
            v[0] = parse_float(&t); v[1] = parse_float(&t); v[2] = parse_float(&t);
        } else if (t[0] == 'v' && t[1] == 't' && OBJ_IS_SPACE(t[2])) {
            t += 3;
            float *v = (float *)grow_push(&T, 1);
            v[0] = parse_float(&t); v[1] = parse_float(&t);
        } else if (t[0] == 'f' && OBJ_IS_SPACE(t[1])) {
            static int cv[OBJ_MAX_CORNERS], cvt[OBJ_MAX_CORNERS], cvn[OBJ_MAX_CORNERS];
            size_t corners = 0;
            t += 2;
            skip_blank(&t);
            while (!OBJ_IS_NEW_LINE(t[0])) {
                if (corners == OBJ_MAX_CORNERS) fatal("a face has too many corners", NULL);
                parse_corner(&t, &cv[corners], &cvt[corners], &cvn[corners]);
                skip_blank_cr(&t);
                corners++;
            }
            /* fan around the first corner; indices are fixed up against the
               element counts at this point of the file */
            size_t tris = 0;
            for (size_t k = 2; k < corners; k++, tris++) {
                const size_t pick[3] = {0, k - 1, k};
                int *f = (int *)grow_push(&F, 3);
                for (int j = 0; j < 3; j++) {
                    f[3 * j + 0] = fix_index(cv[pick[j]], V.n);
                    f[3 * j + 1] = fix_index(cvt[pick[j]], T.n);
                    f[3 * j + 2] = fix_index(cvn[pick[j]], N.n);
                }
            }
            if (tris != 1) nontri++;
            face_lines++;
        } else if ((t[0] == 'g' || t[0] == 'o') && OBJ_IS_SPACE(t[1])) {
            if (face_lines == 0) {
                prev_shape_face_offset = face_lines;
                prev_face_offset = face_lines;
            } else {
                if (shape_idx == 0) {
                    *(unsigned int *)grow_push(&S_off, 1) = 0;
                    *(unsigned int *)grow_push(&S_len, 1) = face_lines - prev_face_offset;
                    shape_idx++;
                    prev_face_offset = face_lines;
                } else if (face_lines - prev_face_offset > 0) {
                    *(unsigned int *)grow_push(&S_off, 1) = prev_face_offset;
                    *(unsigned int *)grow_push(&S_len, 1) = face_lines - prev_face_offset;
                    shape_idx++;
                    prev_face_offset = face_lines;
                }
                prev_shape_face_offset = face_lines;
            }
        }
        /* usemtl / mtllib: material libraries are not used by the renderer */
    }
    free(line);
    if (num_lines == 0) {
        free(V.p); free(N.p); free(T.p); free(F.p); free(S_off.p); free(S_len.p);
        return 2;
    }
    if (face_lines - prev_face_offset > 0 && face_lines - prev_shape_face_offset > 0) {
        *(unsigned int *)grow_push(&S_off, 1) = prev_face_offset;
        *(unsigned int *)grow_push(&S_len, 1) = face_lines - prev_face_offset;
        shape_idx++;
    }

    out->vertices = (float *)V.p;  out->num_vertices = (unsigned int)V.n;
    out->normals = (float *)N.p;   out->num_normals = (unsigned int)N.n;
    out->texcoords = (float *)T.p; out->num_texcoords = (unsigned int)T.n;
    out->faces = (int *)F.p;       out->num_faces = (unsigned int)F.n;
    out->shape_face_offset = (unsigned int *)S_off.p;
    out->shape_length = (unsigned int *)S_len.p;
    out->num_shapes = (unsigned int)shape_idx;
    out->num_face_lines = face_lines;
    out->num_nontriangle_lines = nontri;
    return 0;
}

void rt_obj_free(rt_obj *o)
{
    free(o->vertices); free(o->normals); free(o->texcoords); free(o->faces);
    free(o->shape_face_offset); free(o->shape_length);
    memset(o, 0, sizeof *o);
}

/* ------------------------------------------------------------------------- */
/* JSON (the subset scene files use: objects, arrays, strings, numbers,      */
/* true / false / null)                                                      */
/* ------------------------------------------------------------------------- */
typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } jtype;
typedef struct jval {
    jtype type;
    double num;
    char *str;
    struct jval **items;
    char **keys;
    size_t count;
} jval;

typedef struct { const char *p; } jcur;

static void jskip(jcur *c) { while (*c->p == ' ' || *c->p == '\t' || *c->p == '\n' || *c->p == '\r') c->p++; }
static jval *jparse_value(jcur *c, int depth);

static void jfree(jval *v)
{
    if (!v) return;
    for (size_t i = 0; i < v->count; i++) {
        jfree(v->items[i]);
        if (v->keys) free(v->keys[i]);
    }
    free(v->items); free(v->keys); free(v->str); free(v);
}

static char *jparse_string(jcur *c)
{
    if (*c->p != '"') return NULL;
    c->p++;
    size_t cap = 32, n = 0;
    char *s = (char *)malloc(cap);
    while (*c->p && *c->p != '"') {
        char ch = *c->p++;
        if (ch == '\\') {
            const char e = *c->p++;
            switch (e) {
            case 'n': ch = '\n'; break;
            case 't': ch = '\t'; break;
            case 'r': ch = '\r'; break;
            case 'b': ch = '\b'; break;
            case 'f': ch = '\f'; break;
            case 'u': {                 /* \uXXXX: Latin-1 range kept, the rest becomes '?' */
                unsigned int code = 0;
                for (int k = 0; k < 4; k++) {
                    const char h = *c->p;
                    if (!isxdigit((unsigned char)h)) { free(s); return NULL; }
                    code = code * 16 + (unsigned int)(isdigit((unsigned char)h) ? h - '0' : (tolower(h) - 'a' + 10));
                    c->p++;
                }
                ch = code < 256 ? (char)code : '?';
                break;
            }
            case '\0': free(s); return NULL;
            default: ch = e; break;     /* \" \\ \/ */
            }
        }
        if (n + 2 > cap) { cap *= 2; s = (char *)realloc(s, cap); }
        s[n++] = ch;
    }
    if (*c->p != '"') { free(s); return NULL; }
    c->p++;
    s[n] = '\0';
    return s;
}

static void jappend(jval *v, char *key, jval *item)
{
    v->items = (jval **)realloc(v->items, sizeof(jval *) * (v->count + 1));
    if (v->type == J_OBJ) {
        v->keys = (char **)realloc(v->keys, sizeof(char *) * (v->count + 1));
        v->keys[v->count] = key;
    }
    v->items[v->count++] = item;
}

static jval *jparse_value(jcur *c, int depth)
{
    if (depth > 64) return NULL;
    jskip(c);
    jval *v = (jval *)calloc(1, sizeof(jval));
    const char ch = *c->p;
    if (ch == '{' || ch == '[') {
        const char close = (ch == '{') ? '}' : ']';
        v->type = (ch == '{') ? J_OBJ : J_ARR;
        c->p++;
        jskip(c);
        if (*c->p == close) { c->p++; return v; }
        for (;;) {
            char *key = NULL;
            jskip(c);
            if (v->type == J_OBJ) {
                key = jparse_string(c);
                jskip(c);
                if (!key || *c->p != ':') { free(key); jfree(v); return NULL; }
                c->p++;
            }
            jval *item = jparse_value(c, depth + 1);
            if (!item) { free(key); jfree(v); return NULL; }
            jappend(v, key, item);
            jskip(c);
            if (*c->p == ',') { c->p++; continue; }
            if (*c->p == close) { c->p++; return v; }
            jfree(v);
            return NULL;
        }
    }
    if (ch == '"') {
        v->type = J_STR;
        v->str = jparse_string(c);
        if (!v->str) { jfree(v); return NULL; }
        return v;
    }
    if (!strncmp(c->p, "true", 4)) { v->type = J_BOOL; v->num = 1; c->p += 4; return v; }
    if (!strncmp(c->p, "false", 5)) { v->type = J_BOOL; v->num = 0; c->p += 5; return v; }
    if (!strncmp(c->p, "null", 4)) { v->type = J_NULL; c->p += 4; return v; }
    if (ch == '-' || isdigit((unsigned char)ch)) {
        char *end = NULL;
        v->type = J_NUM;
        v->num = strtod(c->p, &end);
        if (end == c->p) { jfree(v); return NULL; }
        c->p = end;
        return v;
    }
    jfree(v);
    return NULL;
}

static const jval *jget(const jval *o, const char *key)
{
    if (!o || o->type != J_OBJ) return NULL;
    for (size_t i = 0; i < o->count; i++)
        if (!strcmp(o->keys[i], key)) return o->items[i];
    return NULL;
}
/* a missing or non-numeric member reads as 0, as the reference's JSON library does */
static double jnum(const jval *o, const char *key)
{
    const jval *v = jget(o, key);
    return (v && v->type == J_NUM) ? v->num : 0.0;
}
static const char *jstr(const jval *o, const char *key)
{
    const jval *v = jget(o, key);
    return (v && v->type == J_STR) ? v->str : NULL;
}
static const jval *jarr(const jval *o, const char *key)
{
    const jval *v = jget(o, key);
    return (v && v->type == J_ARR) ? v : NULL;
}
static size_t jcount(const jval *a) { return a ? a->count : 0; }
static const jval *jitem(const jval *a, size_t i)
{
    return (a && i < a->count && a->items[i]->type == J_OBJ) ? a->items[i] : NULL;
}

/* ------------------------------------------------------------------------- */
/* 4x4 matrices of the mesh placement: flat arrays, element (i, j) at 4 i + j, */
/* product summed left to right in single precision (libs/vec.h:1240-1262)   */
/* ------------------------------------------------------------------------- */
static void m4_identity(float *m) { memset(m, 0, 16 * sizeof(float)); m[0] = m[5] = m[10] = m[15] = 1.0f; }

static void m4_mul(float *out, const float *a, const float *b)
{
    float o[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            o[4 * i + j] = a[4 * i + 0] * b[0 + j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j] + a[4 * i + 3] * b[12 + j];
    memcpy(out, o, sizeof o);
}

/* unit quaternion {x, y, z, w} -> rotation (libs/vec.h:1162-1198, 1563-1576) */
static void m4_from_quat(float *m, const float *q)
{
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const float xx = qx * x2, xy = qx * y2, xz = qx * z2;
    const float yy = qy * y2, yz = qy * z2, zz = qz * z2;
    const float wx = qw * x2, wy = qw * y2, wz = qw * z2;
    m4_identity(m);
    m[0] = 1.0f - (yy + zz); m[1] = xy - wz;          m[2] = xz + wy;
    m[4] = xy + wz;          m[5] = 1.0f - (xx + zz); m[6] = yz - wx;
    m[8] = xz - wy;          m[9] = yz + wx;          m[10] = 1.0f - (xx + yy);
}

/* ------------------------------------------------------------------------- */
/* scene assembly                                                            */
/* ------------------------------------------------------------------------- */
typedef struct {
    rt_obj obj;
    int mat_index;
    mat4 model;
} mesh_file;

scene *load_scene_json(char *json)
{
    printf("Beginning scene loading...\n");
    jcur cur = {json};
    jval *root = jparse_value(&cur, 0);
    if (!root || root->type != J_OBJ) fatal("the scene description is not valid JSON", NULL);

    scene *s = (scene *)calloc(1, sizeof(scene));
    m4_identity(s->camera_world_matrix);
    printf("Scene name: %s\n", jstr(root, "name") ? jstr(root, "name") : "(null)");

    /* materials, src/loader.c:251-266 */
    {
        const jval *a = jarr(root, "materials");
        s->num_materials = (unsigned int)jcount(a);
        if (s->num_materials == 0) fatal("a scene needs at least one material", NULL);
        s->materials = (material *)calloc(s->num_materials, sizeof(material));
        for (unsigned int i = 0; i < s->num_materials; i++) {
            const jval *m = jitem(a, i);
            s->materials[i].colour[0] = (float)jnum(m, "r");
            s->materials[i].colour[1] = (float)jnum(m, "g");
            s->materials[i].colour[2] = (float)jnum(m, "b");
            s->materials[i].reflectivity = (float)jnum(m, "reflectivity");
        }
        printf("Materials: %d\n", s->num_materials);
    }

    /* spheres and planes, src/loader.c:268-316 */
    {
        const jval *prims = jget(root, "primitives");
        const jval *sa = jarr(prims, "spheres"), *pa = jarr(prims, "planes");
        s->num_spheres = (unsigned int)jcount(sa);
        s->spheres = (sphere *)calloc(s->num_spheres ? s->num_spheres : 1, sizeof(sphere));
        for (unsigned int i = 0; i < s->num_spheres; i++) {
            const jval *o = jitem(sa, i);
            s->spheres[i].pos[0] = (float)jnum(o, "x");
            s->spheres[i].pos[1] = (float)jnum(o, "y");
            s->spheres[i].pos[2] = (float)jnum(o, "z");
            s->spheres[i].radius = (float)jnum(o, "radius");
            s->spheres[i].material_index = (int)jnum(o, "mat_index");
        }
        printf("Spheres: %d\n", s->num_spheres);
        s->num_planes = (unsigned int)jcount(pa);
        s->planes = (plane *)calloc(s->num_planes ? s->num_planes : 1, sizeof(plane));
        for (unsigned int i = 0; i < s->num_planes; i++) {
            const jval *o = jitem(pa, i);
            s->planes[i].pos[0] = (float)jnum(o, "x");
            s->planes[i].pos[1] = (float)jnum(o, "y");
            s->planes[i].pos[2] = (float)jnum(o, "z");
            s->planes[i].norm[0] = (float)jnum(o, "nx");
            s->planes[i].norm[1] = (float)jnum(o, "ny");
            s->planes[i].norm[2] = (float)jnum(o, "nz");
            s->planes[i].material_index = (int)jnum(o, "mat_index");
        }
        printf("Planes: %d\n", s->num_planes);
    }

    /* meshes, src/loader.c:319-438: first every file is parsed and counted,
       then the scene-wide arrays are filled file by file */
    {
        const jval *a = jarr(root, "meshes");
        const size_t files = jcount(a);
        mesh_file *mf = (mesh_file *)calloc(files ? files : 1, sizeof(mesh_file));
        for (size_t i = 0; i < files; i++) {
            const jval *m = jitem(a, i);
            const char *url = jstr(m, "url");
            if (!url) fatal("a mesh entry has no url", NULL);
            long length = 0;
            char *data = load_file(url, &length);
            if (rt_obj_parse(&mf[i].obj, data, (size_t)length) != 0) fatal("Couldn't parse mesh", url);
            free(data);
            if (mf[i].obj.num_nontriangle_lines) fatal("the obj loader only supports triangulated meshes:", url);
            mf[i].mat_index = (int)jnum(m, "mat_index");

            mat4 tr, sc, rot;
            m4_identity(tr); m4_identity(sc); m4_identity(rot);
            if (jget(m, "px")) {
                tr[12] = (float)jnum(m, "px"); tr[13] = (float)jnum(m, "py"); tr[14] = (float)jnum(m, "pz");
            }
            if (jget(m, "sx")) {
                sc[0] = (float)jnum(m, "sx"); sc[5] = (float)jnum(m, "sy"); sc[10] = (float)jnum(m, "sz");
            }
            if (jget(m, "rw")) {
                const float q[4] = {(float)jnum(m, "rx"), (float)jnum(m, "ry"), (float)jnum(m, "rz"), (float)jnum(m, "rw")};
                m4_from_quat(rot, q);
            }
            m4_mul(sc, sc, rot);
            m4_mul(mf[i].model, tr, sc);

            s->num_meshes += mf[i].obj.num_shapes;
            s->num_mesh_verts += mf[i].obj.num_vertices;
            s->num_mesh_nrmls += mf[i].obj.num_normals;
            s->num_mesh_texcoords += mf[i].obj.num_texcoords;
            s->num_mesh_indices += mf[i].obj.num_faces;
        }

        s->meshes = (mesh *)calloc(s->num_meshes ? s->num_meshes : 1, sizeof(mesh));
        s->mesh_verts = (vec3 *)calloc(s->num_mesh_verts ? s->num_mesh_verts : 1, sizeof(vec3));
        s->mesh_nrmls = (vec3 *)calloc(s->num_mesh_nrmls ? s->num_mesh_nrmls : 1, sizeof(vec3));
        s->mesh_texcoords = (vec2 *)calloc(s->num_mesh_texcoords ? s->num_mesh_texcoords : 1, sizeof(vec2));
        s->mesh_indices = (ivec3 *)calloc(s->num_mesh_indices ? s->num_mesh_indices : 1, sizeof(ivec3));

        unsigned int mesh_off = 0, vert_off = 0, nrml_off = 0, tex_off = 0, index_off = 0;
        for (size_t i = 0; i < files; i++) {
            const rt_obj *o = &mf[i].obj;
            /* src/loader.c:140-198 */
            for (unsigned int sh = 0; sh < o->num_shapes; sh++) {
                mesh *m = s->meshes + mesh_off++;
                m->min[0] = m->min[1] = m->min[2] = FLT_MAX;
                m->max[0] = m->max[1] = m->max[2] = -FLT_MAX;
                memcpy(m->model, mf[i].model, sizeof(mat4));
                m->index_offset = (int)index_off;
                m->num_indices = (int)(o->shape_length[sh] * 3u);
                m->material_index = mf[i].mat_index;
                for (unsigned int f = 0; f < o->shape_length[sh]; f++) {
                    for (int j = 0; j < 3; j++) {
                        const int *corner = o->faces + 3 * ((size_t)(f + o->shape_face_offset[sh]) * 3 + j);
                        const int vi = corner[0], vti = corner[1], vni = corner[2];
                        if (vi < 0 || (unsigned int)vi >= o->num_vertices) fatal("a face refers to a vertex that does not exist", NULL);
                        if (index_off >= s->num_mesh_indices) fatal("mesh shapes cover more faces than the file has", NULL);
                        const float *vertex = o->vertices + 3 * (size_t)vi;
                        for (int k = 0; k < 3; k++) {
                            m->min[k] = vertex[k] < m->min[k] ? vertex[k] : m->min[k];
                            m->max[k] = vertex[k] > m->max[k] ? vertex[k] : m->max[k];
                        }
                        int *dst = (int *)s->mesh_indices[index_off];
                        dst[0] = (int)(vert_off + (unsigned int)vi);
                        dst[1] = (int)(nrml_off + (unsigned int)vni);
                        dst[2] = (int)(tex_off + (unsigned int)vti);
                        dst[3] = (int)mesh_off - 1;
                        index_off++;
                    }
                }
            }
            /* src/loader.c:203-225: three floats per element into four-float slots */
            for (unsigned int v = 0; v < o->num_vertices; v++) memcpy(s->mesh_verts + vert_off++, o->vertices + 3 * (size_t)v, 3 * sizeof(float));
            for (unsigned int v = 0; v < o->num_normals; v++) memcpy(s->mesh_nrmls + nrml_off++, o->normals + 3 * (size_t)v, 3 * sizeof(float));
            if (o->num_texcoords) memcpy(s->mesh_texcoords + tex_off, o->texcoords, sizeof(vec2) * (size_t)o->num_texcoords);
            tex_off += o->num_texcoords;
        }
        printf("%i and %i\n", (int)vert_off, (int)s->num_mesh_verts);
        if (mesh_off != s->num_meshes || vert_off != s->num_mesh_verts || nrml_off != s->num_mesh_nrmls ||
            tex_off != s->num_mesh_texcoords || index_off != s->num_mesh_indices)
            fatal("mesh shapes do not cover every face of their file", NULL);
        printf("Meshes: %d\nVertices: %d\nIndices: %d\n", s->num_meshes, s->num_mesh_verts, s->num_mesh_indices);

        for (size_t i = 0; i < files; i++) rt_obj_free(&mf[i].obj);
        free(mf);
    }

    s->materials_changed = s->spheres_changed = s->planes_changed = s->meshes_changed = true;
    printf("Finshed scene loading.\n\n");
    fflush(stdout);
    jfree(root);
    return s;
}

scene *load_scene_json_url(char *url)
{
    long length = 0;
    char *data = load_file(url, &length);
    scene *s = load_scene_json(data);
    free(data);
    return s;
}
