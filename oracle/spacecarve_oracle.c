/*
 * spacecarve_oracle.c -- CPU restatement of the reference's voxel back-projection
 * kernels.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X space-carving engine.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
 * load or call it; nothing under plant-3d-vision_amd/ (the product) does.
 *
 * It restates, in plain C99, the algorithm of
 *   /root/reference/plant3dvision/kernels/backprojection.c  (carve :57-84,
 *       average :36-55, backproject_point :3-34)
 *   /root/reference/plant3dvision/kernels/common.h          (unravel_index :1-12)
 * under the canonical arithmetic SURVEY.md 8c fixes where OpenCL C leaves the
 * result implementation-defined:
 *   (1) IEEE-754 binary32, every multiply and add rounded separately, evaluated
 *       left to right as written (build with -ffp-contract=off, no -ffast-math,
 *       no -march=native: x86-64 SSE2 scalar float, no FMA);
 *   (2) correctly rounded division;
 *   (3) (int) of NaN, +-inf or anything outside int32 gives INT_MIN (x86
 *       cvttss2si) -- written out in cvt_trunc() so it is defined C;
 *   (4) int -> float of a voxel index is exact below 2^24.
 *
 * PIN STATUS: see DESIGN.md "Oracle".  The reference's own tests hold no
 * numeric vectors for carve/average (tests/unit/test_cl.py:5-9 only builds two
 * objects); the conventions they do pin (tests/unit/test_proc3d.py:12-30) are
 * checked in tests/test_oracle.py.  No output of the reference itself exists
 * to compare with (its kernels need OpenCL images, which neither box offers):
 * PARITY UNPINNED against an execution of the reference.
 *
 * Layout: labels/values are C-order [nx][ny][nz], z fastest (common.h:6-8).
 * mask is row-major [H][W]; u indexes columns, v rows (cl.py:217 builds the
 * image from an (H, W) array).
 */
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* x86 cvttss2si: truncation toward zero, INT_MIN when not representable. */
static int cvt_trunc(float f) {
    if (!(f > -2147483904.0f && f < 2147483648.0f)) {
        return INT_MIN; /* NaN, +-inf, |f| >= 2^31 */
    }
    return (int)f;
}

typedef struct {
    int x, y, z;
} int3_t;

/* common.h:1-12 */
static int3_t unravel_index(int64_t idx, const int32_t *shape) {
    int64_t ny = shape[1];
    int64_t nz = shape[2];
    int3_t r;
    int64_t i = idx / (ny * nz);
    int64_t j = (idx - i * ny * nz) / nz;
    int64_t k = idx - i * ny * nz - j * nz;
    r.x = (int)i;
    r.y = (int)j;
    r.z = (int)k;
    return r;
}

/* backprojection.c:3-34.  Returns 1 and (u, v) when the point lands in the image. */
static int backproject_point(float px, float py, float pz, const float *intrinsics,
                             const float *rot, const float *tvec, int W, int H, int *u,
                             int *v) {
    float f_x = intrinsics[0];
    float f_y = intrinsics[1];
    float c_x = intrinsics[2];
    float c_y = intrinsics[3];

    float p_z = rot[6] * px + rot[7] * py + rot[8] * pz + tvec[2]; /* :11 */
    if (p_z < 0) { /* :13 */
        return 0;
    }
    float p_x = rot[0] * px + rot[1] * py + rot[2] * pz + tvec[0]; /* :17 */
    float p_y = rot[3] * px + rot[4] * py + rot[5] * pz + tvec[1]; /* :18 */

    p_x = p_x / p_z * f_x + c_x; /* :20 */
    p_y = p_y / p_z * f_y + c_y; /* :21 */

    *u = cvt_trunc(p_x); /* :23 */
    *v = cvt_trunc(p_y); /* :24 */

    if (*u < 0 || *u > W - 1) { /* :26 */
        return 0;
    }
    if (*v < 0 || *v > H - 1) { /* :29 */
        return 0;
    }
    return 1;
}

typedef struct {
    int mode; /* 0 carve, 1 average */
    void *state;
    const int32_t *shape;
    const float *volinfo;
    const float *K, *R, *t;
    const void *mask;
    int W, H;
    int64_t begin, end;
    /* the state may hold only some x-planes of the grid (a rank's share): local plane p is
     * plane first + p * stride of the grid; begin / end count the elements of the state */
    int64_t first, stride;
} job_t;

/* flat index in the whole grid (what get_global_id(0) is in backprojection.c:59) of element idx of the state */
static int64_t global_index(const job_t *jb, int64_t idx) {
    if (jb->first == 0 && jb->stride == 1) {
        return idx;
    }
    int64_t plane = (int64_t)jb->shape[1] * jb->shape[2];
    int64_t p = idx / plane;
    return (jb->first + p * jb->stride) * plane + (idx - p * plane);
}

/* backprojection.c:57-84 over idx in [begin, end) */
static void carve_range(const job_t *jb) {
    int32_t *labels = (int32_t *)jb->state;
    const int32_t *mask = (const int32_t *)jb->mask;
    const float *vi = jb->volinfo;
    for (int64_t idx = jb->begin; idx < jb->end; ++idx) {
        int3_t ijk = unravel_index(global_index(jb, idx), jb->shape);
        if (labels[idx] == -1) { /* :67 */
            continue;
        }
        float x = vi[0] + ijk.x * vi[3]; /* :71 */
        float y = vi[1] + ijk.y * vi[3]; /* :72 */
        float z = vi[2] + ijk.z * vi[3]; /* :73 */
        int u, v;
        if (!backproject_point(x, y, z, jb->K, jb->R, jb->t, jb->W, jb->H, &u, &v)) {
            continue;
        }
        if (mask[(int64_t)v * jb->W + u] == 0) { /* :79 */
            labels[idx] = -1;
        } else if (labels[idx] == 0) { /* :81 */
            labels[idx] = 1;
        }
    }
}

/* backprojection.c:36-55 over idx in [begin, end) */
static void average_range(const job_t *jb) {
    float *value = (float *)jb->state;
    const float *mask = (const float *)jb->mask;
    const float *vi = jb->volinfo;
    for (int64_t idx = jb->begin; idx < jb->end; ++idx) {
        int3_t ijk = unravel_index(global_index(jb, idx), jb->shape);
        float x = vi[0] + ijk.x * vi[3];
        float y = vi[1] + ijk.y * vi[3];
        float z = vi[2] + ijk.z * vi[3];
        int u, v;
        if (!backproject_point(x, y, z, jb->K, jb->R, jb->t, jb->W, jb->H, &u, &v)) {
            continue;
        }
        /* sampler is LINEAR with integer coordinates (:40-41,54): nearest texel */
        value[idx] += mask[(int64_t)v * jb->W + u];
    }
}

static void *worker(void *arg) {
    const job_t *jb = (const job_t *)arg;
    if (jb->mode == 0) {
        carve_range(jb);
    } else {
        average_range(jb);
    }
    return NULL;
}

static int run_view(int mode, void *state, const int32_t *shape, const float *volinfo,
                    const float *K, const float *R, const float *t, const void *mask, int W,
                    int H, int64_t begin, int64_t end, int nthreads, int64_t first, int64_t stride,
                    int64_t nplanes) {
    if (!state || !shape || !volinfo || !K || !R || !t || !mask) {
        return -1;
    }
    if (shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0 || W <= 0 || H <= 0) {
        return -2;
    }
    if (first < 0 || stride < 1 || nplanes < 0 || (nplanes > 0 && first + (nplanes - 1) * stride >= shape[0])) {
        return -4;
    }
    int64_t n = nplanes * shape[1] * shape[2];
    if (begin < 0) begin = 0;
    if (end < 0 || end > n) end = n;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256];
    pthread_t tids[256];
    int64_t span = end - begin;
    for (int w = 0; w < nthreads; ++w) {
        job_t *jb = &jobs[w];
        jb->mode = mode;
        jb->state = state;
        jb->shape = shape;
        jb->volinfo = volinfo;
        jb->K = K;
        jb->R = R;
        jb->t = t;
        jb->mask = mask;
        jb->W = W;
        jb->H = H;
        jb->begin = begin + span * w / nthreads;
        jb->end = begin + span * (w + 1) / nthreads;
        jb->first = first;
        jb->stride = stride;
    }
    if (nthreads == 1) {
        worker(&jobs[0]);
        return 0;
    }
    for (int w = 0; w < nthreads; ++w) {
        if (pthread_create(&tids[w], NULL, worker, &jobs[w]) != 0) {
            for (int q = 0; q < w; ++q) pthread_join(tids[q], NULL);
            return -3;
        }
    }
    for (int w = 0; w < nthreads; ++w) pthread_join(tids[w], NULL);
    return 0;
}

/*
 * One `carve` launch over the flat voxel range [begin, end) (begin<0 / end<0 = whole grid).
 * labels: int32[nx*ny*nz] read/write; shape: int32[3]; volinfo: {ox, oy, oz, voxel_size}
 * (cl.py:181-187); mask: int32[H*W], the cast cl.py:215 applies.
 */
int oracle_carve_view(int32_t *labels, const int32_t *shape, const float *volinfo,
                      const float *K, const float *R, const float *t, const int32_t *mask,
                      int W, int H, int64_t begin, int64_t end, int nthreads) {
    return run_view(0, labels, shape, volinfo, K, R, t, mask, W, H, begin, end, nthreads, 0, 1, shape ? shape[0] : 0);
}

/*
 * The same launch over a rank's planes only (multi-GPU sharding): labels holds nplanes planes,
 * local plane p being plane first + p * stride of the [nx][ny][nz] grid; every voxel's coordinates come from its
 * index in the WHOLE grid, as get_global_id(0) would give them (backprojection.c:59, 71-73).
 */
int oracle_carve_view_planes(int32_t *labels, const int32_t *shape, const float *volinfo, const float *K,
                             const float *R, const float *t, const int32_t *mask, int W, int H, int64_t first,
                             int64_t stride, int64_t nplanes, int nthreads) {
    return run_view(0, labels, shape, volinfo, K, R, t, mask, W, H, -1, -1, nthreads, first, stride, nplanes);
}

/* One `average` launch; values float32[nx*ny*nz], mask float32[H*W]. */
int oracle_average_view(float *values, const int32_t *shape, const float *volinfo,
                        const float *K, const float *R, const float *t, const float *mask,
                        int W, int H, int64_t begin, int64_t end, int nthreads) {
    return run_view(1, values, shape, volinfo, K, R, t, mask, W, H, begin, end, nthreads, 0, 1, shape ? shape[0] : 0);
}

/*
 * Projection of explicit voxel indices, for edge-case tests: writes u, v (INT_MIN-style
 * raw casts included) and ok[n] = 1 when the reference would touch mask[v][u].
 */
int oracle_project(const int32_t *ijk, int64_t n, const float *volinfo, const float *K,
                   const float *R, const float *t, int W, int H, int32_t *u_out,
                   int32_t *v_out, int32_t *ok) {
    for (int64_t q = 0; q < n; ++q) {
        float x = volinfo[0] + ijk[3 * q + 0] * volinfo[3];
        float y = volinfo[1] + ijk[3 * q + 1] * volinfo[3];
        float z = volinfo[2] + ijk[3 * q + 2] * volinfo[3];
        int u = INT_MIN, v = INT_MIN;
        ok[q] = backproject_point(x, y, z, K, R, t, W, H, &u, &v);
        u_out[q] = u;
        v_out[q] = v;
    }
    return 0;
}

/*
 * The sample generator and result word of the engine's projection self-test
 * (include/spacecarve.h, sc_selftest_project), restated on the reference arithmetic above:
 * sample i draws its pose record from (seed, i / 64) and its voxel from (seed, i); the word is
 * v * W + u + 1 when backproject_point accepts, else 0; a digest is the sum over 65536
 * consecutive samples of mix32(word ^ (uint32)i).
 */
typedef struct {
    float K[4], R[9], t[3];
    float ox, oy, oz, vs;
    int32_t W, H, nx, ny, nz, pad[3];
} pose_rec_t;

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

typedef struct {
    int64_t begin, end;
    uint32_t seed, nposes;
    const pose_rec_t *poses;
    const int32_t *ijk, *pose_idx;
    uint32_t *words;
    uint64_t *digests;
} st_job_t;

static void *selftest_worker(void *arg) {
    const st_job_t *jb = (const st_job_t *)arg;
    for (int64_t i = jb->begin; i < jb->end; ++i) {
        uint32_t p;
        int vi, vj, vk;
        if (jb->ijk) {
            p = jb->pose_idx ? (uint32_t)jb->pose_idx[i] : 0u;
            vi = jb->ijk[3 * i]; vj = jb->ijk[3 * i + 1]; vk = jb->ijk[3 * i + 2];
        } else {
            uint64_t w = (uint64_t)i >> 6;
            uint32_t h = mix32((uint32_t)w ^ jb->seed);
            h = mix32(h + (uint32_t)(w >> 32) * 0x9e3779b9u);
            p = h % jb->nposes;
            uint32_t h2 = mix32((uint32_t)i * 0x9e3779b9u + jb->seed + (uint32_t)((uint64_t)i >> 32));
            uint32_t h3 = mix32(h2 ^ 0x85ebca6bu), h4 = mix32(h3 + 0xc2b2ae35u);
            vi = (int)(h2 % (uint32_t)jb->poses[p].nx);
            vj = (int)(h3 % (uint32_t)jb->poses[p].ny);
            vk = (int)(h4 % (uint32_t)jb->poses[p].nz);
        }
        const pose_rec_t *r = &jb->poses[p];
        float x = r->ox + vi * r->vs; /* backprojection.c:71-73 */
        float y = r->oy + vj * r->vs;
        float z = r->oz + vk * r->vs;
        int u = 0, v = 0;
        uint32_t word = 0;
        if (backproject_point(x, y, z, r->K, r->R, r->t, r->W, r->H, &u, &v)) {
            word = (uint32_t)v * (uint32_t)r->W + (uint32_t)u + 1u;
        }
        if (jb->words) jb->words[i] = word;
        if (jb->digests) jb->digests[i >> 16] += (uint64_t)mix32(word ^ (uint32_t)i);
    }
    return NULL;
}

int oracle_selftest_project(int64_t count, uint32_t seed, int nposes, const float *poses,
                            const int32_t *ijk, const int32_t *pose_idx, uint32_t *words_out,
                            uint64_t *digests_out, int nthreads) {
    if (count < 0 || nposes < 1 || !poses || (!words_out && !digests_out)) return -1;
    int64_t nchunks = (count + 65535) >> 16;
    if (digests_out) memset(digests_out, 0, (size_t)nchunks * sizeof(uint64_t));
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads > nchunks) nthreads = (int)(nchunks > 0 ? nchunks : 1);
    st_job_t jobs[256];
    pthread_t tids[256];
    for (int w = 0; w < nthreads; ++w) { /* whole digest chunks per thread: no shared sums */
        st_job_t *jb = &jobs[w];
        int64_t c0 = nchunks * w / nthreads, c1 = nchunks * (w + 1) / nthreads;
        jb->begin = c0 << 16;
        jb->end = (c1 << 16) < count ? (c1 << 16) : count;
        jb->seed = seed; jb->nposes = (uint32_t)nposes;
        jb->poses = (const pose_rec_t *)poses;
        jb->ijk = ijk; jb->pose_idx = pose_idx;
        jb->words = words_out; jb->digests = digests_out;
    }
    if (nthreads == 1) {
        selftest_worker(&jobs[0]);
        return 0;
    }
    for (int w = 0; w < nthreads; ++w) {
        if (pthread_create(&tids[w], NULL, selftest_worker, &jobs[w]) != 0) {
            for (int q = 0; q < w; ++q) pthread_join(tids[q], NULL);
            return -3;
        }
    }
    for (int w = 0; w < nthreads; ++w) pthread_join(tids[w], NULL);
    return 0;
}

int oracle_abi_version(void) { return 1; }
