// label_points.hip -- SegmentedPointCloud's point back-projection on the GPU.
//
// Replaces the per-point Python loop of plant3dvision/tasks/proc3d.py:203-232 over
// plant3dvision/proc3d.py::backproject_points (:655-659): every point is projected into every
// view in float64 (x = R p + t; x = K x; x /= x_z), the pixel is int(x + 0.5) -- truncation toward
// zero, a DIFFERENT rounding from the carve kernel's -- and, if it is inside the picture, the
// label image's value is added to scores[label][point]; the point's label is the arg-max.
// The reference multiplies with BLAS (`rot @ points.T`), whose summation order is the machine's;
// here each dot product is evaluated left to right without FMA.
#include <hip/hip_runtime.h>


#include <cmath>
#include <cstdint>
#include <cstring>

#include "spacecarve.h"

namespace {

constexpr int kB = 256;
thread_local char g_lerr[256];
int fail_l(int code, const char *msg) {
    strncpy(g_lerr, msg, sizeof g_lerr - 1);
    g_lerr[sizeof g_lerr - 1] = 0;
    return code;
}

__global__ __launch_bounds__(kB) void label_points_kernel(const double *__restrict__ pts, int64_t P,
                                                          int L, int V, const double *__restrict__ K,
                                                          const double *__restrict__ R,
                                                          const double *__restrict__ t,
                                                          const uint8_t *__restrict__ masks, int H, int W,
                                                          double *__restrict__ scores,
                                                          int32_t *__restrict__ labels) {
    int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
    if (i >= P) return;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    for (int l = 0; l < L; ++l) scores[(int64_t)l * P + i] = 0.0;
    for (int v = 0; v < V; ++v) {
        const double *r = R + 9 * v, *k = K + 4 * v, *tv = t + 3 * v;
        // rot @ p + tvec (proc3d.py:656)
        double x0 = ((r[0] * px + r[1] * py) + r[2] * pz) + tv[0];
        double x1 = ((r[3] * px + r[4] * py) + r[5] * pz) + tv[1];
        double x2 = ((r[6] * px + r[7] * py) + r[8] * pz) + tv[2];
        // K @ x with K = [[fx,0,cx],[0,fy,cy],[0,0,1]] (tasks/proc3d.py:221-222, proc3d.py:657)
        double u = ((k[0] * x0 + 0.0 * x1) + k[2] * x2) / x2;  // :658
        double w = ((0.0 * x0 + k[1] * x1) + k[3] * x2) / x2;
        double uf = u + 0.5, wf = w + 0.5;  // tasks/proc3d.py:224
        if (!(fabs(uf) < 9.0e15) || !(fabs(wf) < 9.0e15)) continue;  // NaN/inf: int() is out of range -> outside
        int64_t pu = (int64_t)uf, pv = (int64_t)wf;  // astype(int): toward zero
        if (pu >= 0 && pu < W && pv >= 0 && pv < H) {  // is_in_pict, :185-186
            for (int l = 0; l < L; ++l)
                scores[(int64_t)l * P + i] += (double)masks[(((int64_t)l * V + v) * H + pv) * W + pu];  // :229-230
        }
    }
    int best = 0;  // np.argmax: first maximum (:232)
    double bs = scores[i];
    for (int l = 1; l < L; ++l) {
        double s = scores[(int64_t)l * P + i];
        if (s > bs) { bs = s; best = l; }
    }
    labels[i] = best;
}

}  // namespace

extern "C" {

const char *sc_label_points_last_error(void) { return g_lerr; }

int sc_label_points(const double *points, int64_t P, int L, int V, const double *K, const double *R,
                    const double *t, const void *masks, int masks_on_device, int H, int W, int device,
                    double *scores_out, int32_t *labels_out) {
    if (!points || !K || !R || !t || !masks || !scores_out || !labels_out)
        return fail_l(SC_ERR_INVALID, "null argument");
    if (P < 0 || L <= 0 || V < 0 || H <= 0 || W <= 0) return fail_l(SC_ERR_INVALID, "bad sizes");
    if (P == 0) return SC_OK;
    int rc = SC_OK;
    double *pts_d = nullptr, *K_d = nullptr, *R_d = nullptr, *t_d = nullptr, *sc_d = nullptr;
    int32_t *lab_d = nullptr;
    void *m_d = nullptr;
    const size_t mbytes = (size_t)L * V * H * W;
#define L_TRY(expr)                                                                              \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) { rc = fail_l(_e == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE, hipGetErrorString(_e)); goto done; } \
    } while (0)
    L_TRY(hipSetDevice(device));
    L_TRY(hipMalloc(&pts_d, (size_t)P * 24));
    L_TRY(hipMalloc(&K_d, (size_t)(V ? V : 1) * 32));
    L_TRY(hipMalloc(&R_d, (size_t)(V ? V : 1) * 72));
    L_TRY(hipMalloc(&t_d, (size_t)(V ? V : 1) * 24));
    L_TRY(hipMalloc(&sc_d, (size_t)L * P * 8));
    L_TRY(hipMalloc(&lab_d, (size_t)P * 4));
    L_TRY(hipMemcpy(pts_d, points, (size_t)P * 24, hipMemcpyHostToDevice));
    if (V) {
        L_TRY(hipMemcpy(K_d, K, (size_t)V * 32, hipMemcpyHostToDevice));
        L_TRY(hipMemcpy(R_d, R, (size_t)V * 72, hipMemcpyHostToDevice));
        L_TRY(hipMemcpy(t_d, t, (size_t)V * 24, hipMemcpyHostToDevice));
    }
    if (masks_on_device) {
        m_d = const_cast<void *>(masks);
    } else {
        L_TRY(hipMalloc(&m_d, mbytes ? mbytes : 1));
        if (mbytes) L_TRY(hipMemcpy(m_d, masks, mbytes, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(label_points_kernel, dim3((uint32_t)((P + kB - 1) / kB)), dim3(kB), 0, nullptr, pts_d, P,
                       L, V, K_d, R_d, t_d, static_cast<const uint8_t *>(m_d), H, W, sc_d, lab_d);
    L_TRY(hipGetLastError());
    L_TRY(hipMemcpy(scores_out, sc_d, (size_t)L * P * 8, hipMemcpyDeviceToHost));
    L_TRY(hipMemcpy(labels_out, lab_d, (size_t)P * 4, hipMemcpyDeviceToHost));
done:
    (void)hipDeviceSynchronize();
    if (!masks_on_device && m_d) (void)hipFree(m_d);
    void *bufs[] = {pts_d, K_d, R_d, t_d, sc_d, lab_d};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    return rc;
#undef L_TRY
}

}  // extern "C"
