// kbest_costs.hip -- cost-matrix construction on the device: the producers of the
// k-best path's input, so that a frame's association runs on the GPU from (means, covariances) or bounding boxes
// to probabilities / matches without the cost matrix ever visiting the host.
#include <hip/hip_runtime.h>

#include "kbest_engine.h"

namespace kb {

__device__ __forceinline__ double c_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

// x = S^-1 d for a symmetric 3x3 S by LDL^T with diagonal pivoting (largest remaining |diagonal| first): the
// published algorithm of Eigen::LDLT, which computeQuadricCostMatrix calls (assignment.cpp:717).  Same operation
// order as the checker's restatement, so the two agree bit for bit; against Eigen itself (absent, unpinned) the
// contract is 1e-12 relative.
__device__ void ldlt3_solve(const double *S, const double *d, double *x)
{
    double A[3][3], b[3], L[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, D[3];
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        b[i] = d[i];
#pragma unroll
        for (int j = 0; j < 3; j++) A[i][j] = S[i * 3 + j];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int piv = k;
#pragma unroll
        for (int i = k + 1; i < 3; i++) if (fabs(A[i][i]) > fabs(A[piv][piv])) piv = i;
        if (piv != k) {
#pragma unroll
            for (int j = 0; j < 3; j++) { const double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
#pragma unroll
            for (int i = 0; i < 3; i++) { const double t = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = t; }
#pragma unroll
            for (int j = 0; j < 3; j++) if (j < k) { const double t = L[k][j]; L[k][j] = L[piv][j]; L[piv][j] = t; }
            const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        D[k] = A[k][k];
#pragma unroll
        for (int i = k + 1; i < 3; i++) L[i][k] = A[i][k] / D[k];
#pragma unroll
        for (int i = k + 1; i < 3; i++)
#pragma unroll
            for (int j = k + 1; j < 3; j++) A[i][j] = A[i][j] - L[i][k] * D[k] * L[j][k];
    }
    double y[3], z[3];
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = b[perm[i]];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) if (j < i) y[i] = y[i] - L[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = y[i] / D[i];
#pragma unroll
    for (int i = 2; i >= 0; i--)
#pragma unroll
        for (int j = 0; j < 3; j++) if (j > i) y[i] = y[i] - L[j][i] * y[j];
#pragma unroll
    for (int i = 0; i < 3; i++) z[perm[i]] = y[i];
#pragma unroll
    for (int i = 0; i < 3; i++) x[i] = z[i];
}

// computeQuadricCostMatrix (assignment.cpp:705-722): one workgroup per frame, one thread per (landmark,
// measurement) pair; +inf everywhere else except the gate value on each column's own dummy row (:713-720).
__global__ void __launch_bounds__(256) quadric_cost_kernel(QuadricParams p)
{
    const int b = blockIdx.x;
    const int nL = p.nL[b], nM = p.nM[b], nR = nL + nM;
    const double *m1 = p.landMean + 3 * p.landOff[b], *c1 = p.landCov + 9 * p.landOff[b];
    const double *m2 = p.measMean + 3 * p.measOff[b], *c2 = p.measCov + 9 * p.measOff[b];
    double *out = p.cost + p.costOff[b];
    for (int i = threadIdx.x; i < nR * nM; i += blockDim.x) {
        const int c = i / nR, r = i - c * nR;
        double val = c_inf();
        if (r < nL) {
            double d[3], S[9], x[3];
#pragma unroll
            for (int j = 0; j < 3; j++) d[j] = m1[r * 3 + j] - m2[c * 3 + j];
#pragma unroll
            for (int j = 0; j < 9; j++) S[j] = c1[r * 9 + j] + c2[c * 9 + j];
            ldlt3_solve(S, d, x);
            val = d[0] * x[0] + d[1] * x[1] + d[2] * x[2];
        } else if (r == nL + c) {
            val = p.gate;
        }
        out[i] = val;
    }
}

// boundBox::IoU (boundBox.h:62-75): `a` is *this (its xOffset is applied), `o` is the other box.
__device__ __forceinline__ double bb_iou(const double *a, const double *o)
{
    const double l = fmax(a[0] + a[4], o[0]), r = fmin(a[2] + a[4], o[2]);
    const double t = fmax(a[1], o[1]), bt = fmin(a[3], o[3]);
    if (l >= r || t >= bt) return 0.0;
    const double inter = (r - l) * (bt - t);
    const double areaA = (a[2] - a[0]) * (a[3] - a[1]), areaO = (o[2] - o[0]) * (o[3] - o[1]);
    return inter / (areaA + areaO - inter);
}

// computeBBCostMatrix (assignment.cpp:777-797): rows = right boxes + one dummy per left box, columns = left boxes,
// profits = min of the two (asymmetric) IoUs, -inf fill, gate profit on the dummies.
__global__ void __launch_bounds__(256) bb_cost_kernel(BoxParams p)
{
    const int b = blockIdx.x;
    const int nL = p.nL[b], nR = p.nR[b], nRows = nR + nL;
    const double *L = p.boxL + 5 * p.offL[b], *R = p.boxR + 5 * p.offR[b];
    double *out = p.cost + p.costOff[b];
    for (int i = threadIdx.x; i < nRows * nL; i += blockDim.x) {
        const int c = i / nRows, r = i - c * nRows;
        double val = -c_inf();
        if (r < nR) {
            const double i1 = bb_iou(R + 5 * r, L + 5 * c), i2 = bb_iou(L + 5 * c, R + 5 * r);
            val = i1 < i2 ? i1 : i2;
        } else if (r == nR + c) {
            val = p.gate;
        }
        out[i] = val;
    }
}

// asgnBB epilogue (assignment.cpp:769-773): column c matched to right box row4col[c] unless that is a dummy.
__global__ void __launch_bounds__(64) bb_assign_kernel(BoxParams p, const int *row4col, const int *nf, int k, int maxCol)
{
    const int b = blockIdx.x;
    const int nL = p.nL[b], nR = p.nR[b];
    for (int c = threadIdx.x; c < nL; c += 64) {
        int a = -1;
        if (nf[b] > 0) { const int r = row4col[(long long)b * k * maxCol + c]; if (r < nR) a = r; }
        p.assign[p.offL[b] + c] = a;
    }
}

hipError_t launch_quadric_costs(const QuadricParams &p, int B, hipStream_t stream)
{
    hipLaunchKernelGGL(quadric_cost_kernel, dim3(B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_bb_costs(const BoxParams &p, int B, hipStream_t stream)
{
    hipLaunchKernelGGL(bb_cost_kernel, dim3(B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_bb_assign(const BoxParams &p, const int *row4col, const int *nf, int k, int maxCol, int B,
                            hipStream_t stream)
{
    hipLaunchKernelGGL(bb_assign_kernel, dim3(B), dim3(64), 0, stream, p, row4col, nf, k, maxCol);
    return hipGetLastError();
}


// toProbs (assignment.h:19, assignment.cpp:527-542): m = min of the vector; x -> exp(m - x) where m + 42 > x, else 0.
// One workgroup; in place.  (The reference uses it on the permanent path only; it is elementwise, so this is all of it.)
__global__ void __launch_bounds__(256) to_probs_kernel(double *x, long long n)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double m = c_inf();
    for (long long i = tid; i < n; i += 256) m = x[i] < m ? x[i] : m;  // std::min_element: first minimum, same value
    red[tid] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] = red[tid + w] < red[tid] ? red[tid + w] : red[tid];
        __syncthreads();
    }
    m = red[0];
    const double GATE = 42.0;  // assignment.cpp:9
    for (long long i = tid; i < n; i += 256) {
        const double c = x[i];
        x[i] = (m + GATE > c) ? exp(m - c) : 0.0;  // :536-540
    }
}

hipError_t launch_to_probs(double *x, long long n, hipStream_t stream)
{
    hipLaunchKernelGGL(to_probs_kernel, dim3(1), dim3(256), 0, stream, x, n);
    return hipGetLastError();
}

}  // namespace kb
