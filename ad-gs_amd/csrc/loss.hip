// Fused L1 + SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward and backward for gfx950.
// Reference: utils/loss_utils.py:20-68 (called at train.py:79-80).  One 32x16 output tile per workgroup:
// the 42x26 input halo is staged in LDS, the separable window runs as a horizontal pass (into LDS) and a
// vertical pass, both as SLIDING WINDOWS IN REGISTERS: a thread forms 4 neighbouring outputs of a row from
// 14 inputs read as four 16-byte LDS words (instead of 4 x 11 scalar reads), and 2 outputs of a column from
// 12 reads (instead of 22).  The 16x16-tile, one-output-per-thread form of round 2 was LDS-bound (11 reads
// per output and pass, 2-way bank conflicts, one L2 round trip per staged element): 129 + 93 us per C3 frame;
// now 99 + 75.  The arithmetic per output (tap order, fused multiply-adds) is unchanged.
#include "common.h"
#include "../../include/adgs_loss.h"
#include <algorithm>

namespace adgs {
namespace {

constexpr int TSX = 32, TSY = 16;         // output tile
constexpr int WR = 5;                     // window radius (11 taps)
constexpr int NT = 2 * WR + 1;
constexpr int HSX = TSX + 2 * WR, HSY = TSY + 2 * WR;      // halo 42 x 26
constexpr int SSTR = 44;                  // floats per staged row (16-byte aligned; 4 sx + 15 <= 43)
constexpr int HSTR = TSX + 4;             // floats per row of the horizontally filtered images
constexpr int LT = 256;                   // threads per workgroup
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

// gaussian(11, 1.5) of utils/loss_utils.py:26-28: exp(-(x-5)^2 / (2 sigma^2)) normalised by the sum (float32)
struct Window { float g[NT]; };
static Window make_window() {
	Window w; float s = 0.f;
	for (int x = 0; x < 2 * WR + 1; x++) { w.g[x] = (float)std::exp(-(double)((x - WR) * (x - WR)) / (2.0 * 1.5 * 1.5)); s += w.g[x]; }
	for (int x = 0; x < 2 * WR + 1; x++) w.g[x] = w.g[x] / s;
	return w;
}

// Staging of NA halo images at once: zero padding outside the image (F.conv2d padding=5).  All loads of a thread (5 per image, at
// clamped addresses, unconditional) are issued before the first LDS store: a rolled loop of conditional loads paid one L2 round
// trip per element and image -- the staging, not the window arithmetic, was what these kernels' time went into.
template <int NA>
__device__ __forceinline__ void stage_halos(float (*const (&s)[NA])[SSTR], const float* const (&src)[NA], size_t plane, int x0, int y0, int H, int W, int tid) {
	constexpr int NIT = (HSY * HSX + LT - 1) / LT;
	float v[NA][NIT];
#pragma unroll
	for (int it = 0; it < NIT; it++) {
		const int i = min(tid + it * LT, HSY * HSX - 1);
		const int ly = i / HSX, lx = i - ly * HSX;
		const int gy = min(max(y0 + ly - WR, 0), H - 1), gx = min(max(x0 + lx - WR, 0), W - 1);
#pragma unroll
		for (int a = 0; a < NA; a++) v[a][it] = src[a][plane + (size_t)gy * W + gx];
	}
#pragma unroll
	for (int it = 0; it < NIT; it++) {
		const int i = tid + it * LT;
		if (i >= HSY * HSX) break;
		const int ly = i / HSX, lx = i - ly * HSX, gy = y0 + ly - WR, gx = x0 + lx - WR;
		const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
		for (int a = 0; a < NA; a++) s[a][ly][lx] = in ? v[a][it] : 0.f;
	}
}
// 16 consecutive floats of a staged row (14 are used) as four 16-byte LDS reads
__device__ __forceinline__ void load_run(const float* row, float (&u)[16]) {
	const float4* r = reinterpret_cast<const float4*>(row);
#pragma unroll
	for (int q = 0; q < 4; q++) { const float4 v = r[q]; u[4 * q] = v.x; u[4 * q + 1] = v.y; u[4 * q + 2] = v.z; u[4 * q + 3] = v.w; }
}
__device__ __forceinline__ float4 window4(const Window& win, const float (&u)[16]) {
	float o[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		float a = 0.f;
#pragma unroll
		for (int k = 0; k < NT; k++) a += win.g[k] * u[j + k];
		o[j] = a;
	}
	return make_float4(o[0], o[1], o[2], o[3]);
}

__global__ void __launch_bounds__(LT) l1_ssim_fwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, Window win,
	double* __restrict__ sums, float* __restrict__ d_mu1, float* __restrict__ d_e11, float* __restrict__ d_e12) {
	__shared__ __attribute__((aligned(16))) float s1[HSY][SSTR], s2[HSY][SSTR];
	__shared__ __attribute__((aligned(16))) float h[5][HSY][HSTR];      // horizontally filtered x1, x2, x1^2, x2^2, x1 x2
	__shared__ double red[2][LT / WAVE];
	const int tid = threadIdx.x;
	const int x0 = blockIdx.x * TSX, y0 = blockIdx.y * TSY;
	const size_t plane = (size_t)blockIdx.z * H * W;
	{
		float (*const dst[2])[SSTR] = { s1, s2 };
		const float* const src[2] = { img, gt };
		stage_halos<2>(dst, src, plane, x0, y0, H, W, tid);
	}
	__syncthreads();
	{	// horizontal pass: row r, outputs 4 sx .. 4 sx + 3
		const int r = tid >> 3, sx = tid & 7;
		if (r < HSY) {
			float u[16], v[16], t[16];
			load_run(&s1[r][4 * sx], u); load_run(&s2[r][4 * sx], v);
			*reinterpret_cast<float4*>(&h[0][r][4 * sx]) = window4(win, u);
			*reinterpret_cast<float4*>(&h[1][r][4 * sx]) = window4(win, v);
#pragma unroll
			for (int i = 0; i < 16; i++) t[i] = u[i] * u[i];
			*reinterpret_cast<float4*>(&h[2][r][4 * sx]) = window4(win, t);
#pragma unroll
			for (int i = 0; i < 16; i++) t[i] = v[i] * v[i];
			*reinterpret_cast<float4*>(&h[3][r][4 * sx]) = window4(win, t);
#pragma unroll
			for (int i = 0; i < 16; i++) t[i] = u[i] * v[i];
			*reinterpret_cast<float4*>(&h[4][r][4 * sx]) = window4(win, t);
		}
	}
	__syncthreads();
	// vertical pass: column tx, outputs rows 2 g and 2 g + 1
	const int tx = tid & (TSX - 1), g = tid >> 5;
	float acc[5][2];
#pragma unroll
	for (int q = 0; q < 5; q++) {
		float c[NT + 1];
#pragma unroll
		for (int j = 0; j < NT + 1; j++) c[j] = h[q][2 * g + j][tx];
#pragma unroll
		for (int o = 0; o < 2; o++) {
			float a = 0.f;
#pragma unroll
			for (int k = 0; k < NT; k++) a += win.g[k] * c[o + k];
			acc[q][o] = a;
		}
	}
	double l1 = 0.0, sm = 0.0;
#pragma unroll
	for (int o = 0; o < 2; o++) {
		const int ty = 2 * g + o, gx = x0 + tx, gy = y0 + ty;
		if (gx >= W || gy >= H) continue;
		const float mu1 = acc[0][o], mu2 = acc[1][o], e11 = acc[2][o], e22 = acc[3][o], e12 = acc[4][o];
		const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
		const float sg1 = e11 - mu1_sq, sg2 = e22 - mu2_sq, sg12 = e12 - mu12;
		const float A1 = 2.f * mu12 + C1, A2 = 2.f * sg12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = sg1 + sg2 + C2;
		const float D = B1 * B2, inv = 1.f / D;
		sm += (double)((A1 * A2) * inv);
		l1 += (double)fabsf(s1[ty + WR][tx + WR] - s2[ty + WR][tx + WR]);
		if (d_mu1) {
			const size_t oo = plane + (size_t)gy * W + gx;
			// partial derivatives of the map w.r.t. the window means mu1, E[x1^2], E[x1 x2] (mu2, E[x2^2] belong to gt)
			const float num = A1 * A2;
			const float dnum = 2.f * mu2 * A2 - 2.f * mu2 * A1;           // dA1 = 2 mu2, dA2 = -2 mu2
			const float dden = 2.f * mu1 * B2 - 2.f * mu1 * B1;           // dB1 = 2 mu1, dB2 = -2 mu1
			d_mu1[oo] = (dnum * D - num * dden) * (inv * inv);
			d_e11[oo] = -num * inv / B2;                                  // dB2 = 1
			d_e12[oo] = 2.f * A1 * inv;                                   // dA2 = 2
		}
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) { l1 += __shfl_xor(l1, off, WAVE); sm += __shfl_xor(sm, off, WAVE); }
	if ((tid & (WAVE - 1)) == 0) { red[0][tid / WAVE] = l1; red[1][tid / WAVE] = sm; }
	__syncthreads();
	if (tid < 2) {
		double t = 0.0;
		for (int w = 0; w < LT / WAVE; w++) t += red[tid][w];
		const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
		atomicAdd(sums + 2 * (b % ADGS_LOSS_SLOTS) + tid, t);
	}
}

__global__ void __launch_bounds__(LT) l1_ssim_bwd_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, Window win,
	const float* __restrict__ d_mu1, const float* __restrict__ d_e11, const float* __restrict__ d_e12,
	const float* __restrict__ g_l1, const float* __restrict__ g_ssim, float inv_n, float* __restrict__ out) {
	__shared__ __attribute__((aligned(16))) float s[3][HSY][SSTR];
	__shared__ __attribute__((aligned(16))) float h[3][HSY][HSTR];
	const int tid = threadIdx.x;
	const int x0 = blockIdx.x * TSX, y0 = blockIdx.y * TSY;
	const size_t plane = (size_t)blockIdx.z * H * W;
	// map pixels outside the image do not exist: contribute 0
	{
		float (*const dst[3])[SSTR] = { s[0], s[1], s[2] };
		const float* const src[3] = { d_mu1, d_e11, d_e12 };
		stage_halos<3>(dst, src, plane, x0, y0, H, W, tid);
	}
	__syncthreads();
	{
		const int r = tid >> 3, sx = tid & 7;
		if (r < HSY) {
#pragma unroll
			for (int q = 0; q < 3; q++) {
				float u[16];
				load_run(&s[q][r][4 * sx], u);
				*reinterpret_cast<float4*>(&h[q][r][4 * sx]) = window4(win, u);
			}
		}
	}
	__syncthreads();
	const int tx = tid & (TSX - 1), g = tid >> 5;
	float acc[3][2];
#pragma unroll
	for (int q = 0; q < 3; q++) {
		float c[NT + 1];
#pragma unroll
		for (int j = 0; j < NT + 1; j++) c[j] = h[q][2 * g + j][tx];
#pragma unroll
		for (int o = 0; o < 2; o++) {
			float a = 0.f;
#pragma unroll
			for (int k = 0; k < NT; k++) a += win.g[k] * c[o + k];
			acc[q][o] = a;
		}
	}
	const float gs = g_ssim ? g_ssim[0] : 0.f, gl = g_l1 ? g_l1[0] : 0.f;
#pragma unroll
	for (int o = 0; o < 2; o++) {
		const int gx = x0 + tx, gy = y0 + 2 * g + o;
		if (gx >= W || gy >= H) continue;
		const size_t oo = plane + (size_t)gy * W + gx;
		const float x1 = img[oo], x2 = gt[oo];
		const float dx = x1 - x2;
		const float sgn = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f);             // torch.abs backward: sign(), 0 at 0
		out[oo] = gl * sgn * inv_n + gs * inv_n * (acc[0][o] + 2.f * x1 * acc[1][o] + x2 * acc[2][o]);
	}
}

// ---------------------------------------------------------------- scale/shift-invariant depth loss
// work layout (doubles): [slot][8] partial sums (a00, a01, a11, b0, b1 of pass 1; L1, A = sum m sgn p, B = sum m sgn of pass 2),
// then 16 scalars: [0..4] the five totals, [5] s, [6] t, [7] det, [8] L1 total, [9] A, [10] B.
constexpr int DSLOTS = 256;
struct DepthScalars { double a00, a01, a11, b0, b1, s, t, det; };

__global__ void __launch_bounds__(256) depth_sums_kernel(int n, const float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ m, double* __restrict__ work) {
	double a00 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const double w = m ? (double)m[i] : 1.0, x = p[i], y = g[i];
		a00 += w * x * x; a01 += w * x; a11 += w; b0 += w * x * y; b1 += w * y;
	}
	double v[5] = { a00, a01, a11, b0, b1 };
#pragma unroll
	for (int q = 0; q < 5; q++) {
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, WAVE);
		if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(work + (size_t)((blockIdx.x * 4 + threadIdx.x / WAVE) % DSLOTS) * 8 + q, v[q]);
	}
}
// one block: totals of the slots -> scale, shift (depth_utils.py:30-45)
__global__ void __launch_bounds__(256) depth_solve_kernel(double* __restrict__ work) {
	__shared__ double s[5][256 / WAVE];
	double* out = work + (size_t)DSLOTS * 8;
	double v[5];
#pragma unroll
	for (int q = 0; q < 5; q++) {
		v[q] = work[(size_t)threadIdx.x * 8 + q];
		work[(size_t)threadIdx.x * 8 + q] = 0.0;      // consumed (see aux_finish_kernel)
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, WAVE);
		if ((threadIdx.x & (WAVE - 1)) == 0) s[q][threadIdx.x / WAVE] = v[q];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		double t[5];
		for (int q = 0; q < 5; q++) { t[q] = 0; for (int w = 0; w < 256 / WAVE; w++) t[q] += s[q][w]; out[q] = t[q]; }
		// the reference forms the sums and the determinant in fp32 (torch.sum of fp32 tensors): round the totals first
		const float a00 = (float)t[0], a01 = (float)t[1], a11 = (float)t[2], b0 = (float)t[3], b1 = (float)t[4];
		const float det = a00 * a11 - a01 * a01;
		float sc = 0.f, sh = 0.f;
		if (det != 0.f) { sc = (a11 * b0 - a01 * b1) / det; sh = (-a01 * b0 + a00 * b1) / det; }
		out[5] = sc; out[6] = sh; out[7] = det;
	}
}
__global__ void __launch_bounds__(256) depth_l1_kernel(int n, const float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ m, double* __restrict__ work) {
	const double* sc = work + (size_t)DSLOTS * 8;
	const float s = (float)sc[5], t = (float)sc[6];
	double L = 0, A = 0, B = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const float w = m ? m[i] : 1.f, x = p[i];
		const float d = (s * x + t) - g[i];
		const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
		L += (double)(fabsf(d) * w); A += (double)(w * sg * x); B += (double)(w * sg);
	}
	double v[3] = { L, A, B };
#pragma unroll
	for (int q = 0; q < 3; q++) {
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, WAVE);
		if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(work + (size_t)((blockIdx.x * 4 + threadIdx.x / WAVE) % DSLOTS) * 8 + 5 + q, v[q]);
	}
}
__global__ void __launch_bounds__(256) depth_finish_kernel(double* __restrict__ work, float* __restrict__ loss) {
	__shared__ double s[3][256 / WAVE];
	double* out = work + (size_t)DSLOTS * 8;
	double v[3];
#pragma unroll
	for (int q = 0; q < 3; q++) {
		v[q] = work[(size_t)threadIdx.x * 8 + 5 + q];
		work[(size_t)threadIdx.x * 8 + 5 + q] = 0.0;
#pragma unroll
		for (int off = WAVE / 2; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, WAVE);
		if ((threadIdx.x & (WAVE - 1)) == 0) s[q][threadIdx.x / WAVE] = v[q];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int q = 0; q < 3; q++) { double t = 0; for (int w = 0; w < 256 / WAVE; w++) t += s[q][w]; out[8 + q] = t; }
		loss[0] = (float)(out[8] / out[2]);                       // sum(|.| m) / sum(m)
	}
}
__global__ void __launch_bounds__(256) depth_bwd_kernel(int n, const float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ m,
	const double* __restrict__ work, const float* __restrict__ g_loss, float* __restrict__ out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double* sc = work + (size_t)DSLOTS * 8;
	const double a00 = sc[0], a01 = sc[1], a11 = sc[2], b0 = sc[3], b1 = sc[4], s = sc[5], t = sc[6], det = sc[7], A = sc[9], B = sc[10];
	const float w = m ? m[i] : 1.f, x = p[i], y = g[i];
	const float d = ((float)s * x + (float)t) - y;
	const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
	double grad = (double)sg * s;
	if (det != 0.0) {
		// through the least-squares solution: s, t depend on a00 = sum m p^2, a01 = sum m p, b0 = sum m p g
		const double s_a00 = -s * a11 / det, s_a01 = (-b1 + 2.0 * a01 * s) / det, s_b0 = a11 / det;
		const double t_a00 = (b1 - t * a11) / det, t_a01 = (-b0 + 2.0 * a01 * t) / det, t_b0 = -a01 / det;
		grad += (A * s_a00 + B * t_a00) * 2.0 * x + (A * s_a01 + B * t_a01) + (A * s_b0 + B * t_b0) * y;
	} else grad = 0.0;                                             // the reference returns python floats (0.0, 0.0): no graph
	out[i] = (float)((double)g_loss[0] * (double)w * grad / a11);
	(void)a00;
}

// ---------------------------------------------------------------- flow re-projection loss and clipped BCE
// Pixel-wise losses of train.py:88-103 that the reference builds from nonzero() (a host synchronisation), gathers and half a
// dozen elementwise kernels each.  Here: one reduction pass (double partial sums in AUX_SLOTS slots) + a one-thread finish,
// and one elementwise backward.  aux work layout (doubles): [slot][2] = (sum, count), then [2*AUX_SLOTS] = sum, [+1] = count.
constexpr int AUX_SLOTS = 256;
struct FlowCam { float M[9]; float KT[3]; float dist; };          // M = K R, KT = K T (3x3 row-major products formed on the host in fp32)
// The camera of the flow target as DEVICE pointers (the reference keeps K / R / T on the GPU, train.py:68-71): nothing is read back
// and nothing is cached on the host.  K == nullptr: the by-value FlowCam is used.
struct FlowCamDev { const float* K; const float* R; const float* T; };
// the same products, in the same order and without contraction, as make_flow_cam forms on the host: both entry points agree bit for bit
__device__ __forceinline__ FlowCam flow_cam_load(const FlowCam& host, const FlowCamDev& d) {
	if (!d.K) return host;
	FlowCam c;
#pragma unroll
	for (int i = 0; i < 3; i++) {
#pragma unroll
		for (int j = 0; j < 3; j++)
			c.M[3 * i + j] = __fadd_rn(__fadd_rn(__fmul_rn(d.K[3 * i], d.R[j]), __fmul_rn(d.K[3 * i + 1], d.R[3 + j])), __fmul_rn(d.K[3 * i + 2], d.R[6 + j]));
		c.KT[i] = __fadd_rn(__fadd_rn(__fmul_rn(d.K[3 * i], d.T[0]), __fmul_rn(d.K[3 * i + 1], d.T[1])), __fmul_rn(d.K[3 * i + 2], d.T[2]));
	}
	c.dist = host.dist;
	return c;
}

struct FlowPix { bool sel; float w, u, v, z, px, py; bool front; };
__device__ __forceinline__ FlowPix flow_pixel(int i, int HW, int W, int H, const float* __restrict__ f, const float* __restrict__ fl,
	const float* __restrict__ vis, const float* __restrict__ op, const FlowCam& c) {
	FlowPix r;
	const float t0 = fl[i], t1 = fl[HW + i];
	r.sel = (vis[i] > 0.5f) && (t0 <= (float)W - 1.0f) && (t0 >= 0.f) && (t1 <= (float)H - 1.0f) && (t1 >= 0.f);       // loss_utils.py:90
	r.w = 0.f; r.u = r.v = r.z = r.px = r.py = 0.f; r.front = false;
	if (!r.sel) return r;
	const float x = f[i], y = f[HW + i], z = f[2 * HW + i];
	r.px = c.M[0] * x + c.M[1] * y + c.M[2] * z + c.KT[0];
	r.py = c.M[3] * x + c.M[4] * y + c.M[5] * z + c.KT[1];
	const float pz = c.M[6] * x + c.M[7] * y + c.M[8] * z + c.KT[2];
	r.front = pz > c.dist;                                         // flow_utils.py:8
	r.z = fmaxf(pz, c.dist);                                       // clamp_min, :9
	r.u = r.px / r.z; r.v = r.py / r.z;
	r.w = (op ? op[i] : 1.f) * (r.front ? 1.f : 0.f);
	return r;
}
__global__ void __launch_bounds__(256) flow_loss_sum_kernel(int H, int W, const float* __restrict__ f, const float* __restrict__ fl,
	const float* __restrict__ vis, const float* __restrict__ op, FlowCam c0, FlowCamDev cd, double* __restrict__ work) {
	const int HW = H * W;
	const FlowCam c = flow_cam_load(c0, cd);
	double sum = 0, cnt = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
		const FlowPix p = flow_pixel(i, HW, W, H, f, fl, vis, op, c);
		if (p.sel) {
			cnt += 1.0;
			sum += (double)((fabsf(p.u - fl[i]) / (float)W + fabsf(p.v - fl[HW + i]) / (float)H) * p.w);      // :103-105
		}
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, WAVE); cnt += __shfl_xor(cnt, off, WAVE); }
	if ((threadIdx.x & (WAVE - 1)) == 0) {
		const size_t slot = (size_t)((blockIdx.x * 4 + threadIdx.x / WAVE) % AUX_SLOTS) * 2;
		atomicAdd(work + slot, sum); atomicAdd(work + slot + 1, cnt);
	}
}
__global__ void __launch_bounds__(256) aux_finish_kernel(double* __restrict__ work, float* __restrict__ loss) {
	__shared__ double s[2][256 / WAVE];
	double a = work[(size_t)threadIdx.x * 2], b = work[(size_t)threadIdx.x * 2 + 1];
	work[(size_t)threadIdx.x * 2] = 0.0; work[(size_t)threadIdx.x * 2 + 1] = 0.0;      // consumed: the slot region is zero again for the next call on this buffer (include/adgs_loss.h)
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) { a += __shfl_xor(a, off, WAVE); b += __shfl_xor(b, off, WAVE); }
	if ((threadIdx.x & (WAVE - 1)) == 0) { s[0][threadIdx.x / WAVE] = a; s[1][threadIdx.x / WAVE] = b; }
	__syncthreads();
	if (threadIdx.x == 0) {
		double ta = 0, tb = 0;
		for (int w = 0; w < 256 / WAVE; w++) { ta += s[0][w]; tb += s[1][w]; }
		work[2 * AUX_SLOTS] = ta; work[2 * AUX_SLOTS + 1] = tb;
		loss[0] = tb > 0 ? (float)(ta / tb) : 0.f;                   // mean over the selected pixels; 0.0 when none (:92-93)
	}
}
__global__ void __launch_bounds__(256) flow_loss_bwd_kernel(int H, int W, const float* __restrict__ f, const float* __restrict__ fl,
	const float* __restrict__ vis, const float* __restrict__ op, FlowCam c0, FlowCamDev cd, const double* __restrict__ work, const float* __restrict__ g_loss,
	float* __restrict__ g_f, float* __restrict__ g_op) {
	const int HW = H * W;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= HW) return;
	const FlowCam c = flow_cam_load(c0, cd);
	const double n = work[2 * AUX_SLOTS + 1];
	const FlowPix p = flow_pixel(i, HW, W, H, f, fl, vis, op, c);
	float gx = 0.f, gy = 0.f, gz = 0.f, go = 0.f;
	if (p.sel && n > 0) {
		const float gl = (float)((double)g_loss[0] / n);
		const float du = p.u - fl[i], dv = p.v - fl[HW + i];
		go = gl * (fabsf(du) / (float)W + fabsf(dv) / (float)H) * (p.front ? 1.f : 0.f);
		const float su = du > 0.f ? 1.f : (du < 0.f ? -1.f : 0.f), sv = dv > 0.f ? 1.f : (dv < 0.f ? -1.f : 0.f);
		const float gu = gl * su * p.w / (float)W, gv = gl * sv * p.w / (float)H;
		// u = px / z, v = py / z; z = max(pz, dist) passes the gradient where pz > dist (the weight is 0 elsewhere)
		const float gpx = gu / p.z, gpy = gv / p.z, gpz = -(gu * p.px + gv * p.py) / (p.z * p.z);
		gx = c.M[0] * gpx + c.M[3] * gpy + c.M[6] * gpz;
		gy = c.M[1] * gpx + c.M[4] * gpy + c.M[7] * gpz;
		gz = c.M[2] * gpx + c.M[5] * gpy + c.M[8] * gpz;
	}
	g_f[i] = gx; g_f[HW + i] = gy; g_f[2 * HW + i] = gz;
	if (g_op) g_op[i] = go;
}

// mean BCE of q = clip(pred, lo, hi) (or 1 - clip) against t = target (or target > 0): train.py:95-103
__device__ __forceinline__ float bce_target(float t, int positive) { return positive ? (t > 0.f ? 1.f : 0.f) : t; }
__global__ void __launch_bounds__(256) bce_clip_sum_kernel(int n, const float* __restrict__ pred, const float* __restrict__ target, float lo, float hi,
	int invert, int positive, double* __restrict__ work) {
	double sum = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const float c = fminf(fmaxf(pred[i], lo), hi);
		const float q = invert ? 1.0f - c : c;
		const float t = bce_target(target[i], positive);
		sum += (double)(-(t * fmaxf(logf(q), -100.f) + (1.f - t) * fmaxf(logf(1.f - q), -100.f)));     // torch clamps the logs at -100
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, WAVE);
	if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(work + (size_t)((blockIdx.x * 4 + threadIdx.x / WAVE) % AUX_SLOTS) * 2, sum);
	if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(work + 1, (double)n);      // the "count" of the shared finish kernel
}
__global__ void __launch_bounds__(256) bce_clip_bwd_kernel(int n, const float* __restrict__ pred, const float* __restrict__ target, float lo, float hi,
	int invert, int positive, const float* __restrict__ g_loss, float* __restrict__ out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float x = pred[i];
	const float c = fminf(fmaxf(x, lo), hi);
	const float q = invert ? 1.0f - c : c;
	const float t = bce_target(target[i], positive);
	// d BCE / dq = (q - t) / (q (1 - q)) with torch's eps-clamped denominator; clamp passes the gradient on [lo, hi]
	const float dq = (q - t) / fmaxf(q * (1.f - q), 1e-12f);
	const bool inside = x >= lo && x <= hi;
	out[i] = inside ? g_loss[0] * (invert ? -dq : dq) / (float)n : 0.f;
}

} // namespace
} // namespace adgs

using namespace adgs;

// ---------------------------------------------------------------- neighbourhood regularisers (train.py:104-113)
//   reg_loss       = mean(sum(var(xyz_deform_param[obj_near_idx], dim=1), dim=-1))     rows of D = 3 C floats, inner = C
//   reg_sigma_loss = mean(sum(var(gs_time_sigma[obj_near_idx], dim=1), dim=-1))        rows of D = 2 floats,   inner = 2
//   sigma_loss     = mean(|frame_gap / mean(exp(gs_time_sigma), dim=-1)|)
// The reference gathers a [G, K, 3, C] copy of the rows (43 MB at C3), runs torch.var / sum / mean over it and scatters the
// gradient back through index_put.  Here one thread owns one (group, column) pair: it reads its K values straight from the K
// gathered rows (consecutive threads = consecutive columns of a row: coalesced), forms the unbiased variance in registers, and
// the backward adds 2 (x - mean) / ((K - 1) denom) to the K rows with one atomic each (a Gaussian can sit in several groups).
// Row indices follow the reference's fancy indexing (`param[obj_near_idx]`): a negative index counts from the end; anything outside
// [-N, N) raises IndexError there.  A kernel cannot raise without a host round trip, so an out-of-range index (a stale obj_near_idx from
// before a prune) makes the LOSS NaN -- and no out-of-bounds access -- and the rows of the affected GROUP receive no gradient (a NaN mean would
// otherwise reach every valid row of the group and, one Adam step later, the parameters and both moments for good).  The host wrapper
// (adgs.loss._GroupVar) validates every new index tensor once and raises IndexError like the reference.
constexpr int GV_MAXK = 32;
__device__ __forceinline__ long long gv_row(long long i, int N) { if (i < 0) i += N; return (i < 0 || i >= N) ? -1 : i; }
__global__ void __launch_bounds__(256) group_var_sum_kernel(int N, int G, int K, int D, const float* __restrict__ x, const long long* __restrict__ idx,
	double denom, double* __restrict__ work) {
	const long long total = (long long)G * D;
	double sum = 0;
	for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
		const int g = (int)(t / D), d = (int)(t % D);
		float v[GV_MAXK], mean = 0.f;
		for (int k = 0; k < K; k++) { const long long r = gv_row(idx[(size_t)g * K + k], N); v[k] = r < 0 ? __int_as_float(0x7fc00000) : x[(size_t)r * D + d]; mean += v[k]; }
		mean /= (float)K;
		float ss = 0.f;
		for (int k = 0; k < K; k++) { const float c = v[k] - mean; ss += c * c; }
		sum += (double)(ss / (float)(K - 1));
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, WAVE);
	if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(work + (size_t)((blockIdx.x * 4 + threadIdx.x / WAVE) % AUX_SLOTS) * 2, sum);
	if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(work + 1, denom);      // the "count" of the shared finish kernel
}
__global__ void __launch_bounds__(256) group_var_bwd_kernel(int N, int G, int K, int D, const float* __restrict__ x, const long long* __restrict__ idx,
	float scale, const float* __restrict__ g_loss, float* __restrict__ dx) {
	const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (long long)G * D) return;
	const int g = (int)(t / D), d = (int)(t % D);
	float v[GV_MAXK], mean = 0.f; long long row[GV_MAXK];
	for (int k = 0; k < K; k++) { row[k] = gv_row(idx[(size_t)g * K + k], N); v[k] = row[k] < 0 ? __int_as_float(0x7fc00000) : x[(size_t)row[k] * D + d]; mean += v[k]; }
	mean /= (float)K;
	const float c = scale * g_loss[0];
	bool valid = true;
	for (int k = 0; k < K; k++) valid = valid && row[k] >= 0;
	if (!valid) return;          // a group with an out-of-range index has no defined variance: its rows receive NO gradient (never NaN) -- the forward's loss is NaN
	for (int k = 0; k < K; k++) atomicAdd(dx + (size_t)row[k] * D + d, c * (v[k] - mean));
}
__global__ void __launch_bounds__(256) sigma_loss_sum_kernel(int N, const float* __restrict__ ls, float gap, double* __restrict__ work) {
	double sum = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
		const float m = 0.5f * (expf(ls[2 * i]) + expf(ls[2 * i + 1]));
		sum += (double)fabsf(gap / m);
	}
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, WAVE);
	if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(work + (size_t)((blockIdx.x * 4 + threadIdx.x / WAVE) % AUX_SLOTS) * 2, sum);
	if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(work + 1, (double)N);
}
__global__ void __launch_bounds__(256) sigma_loss_bwd_kernel(int N, const float* __restrict__ ls, float gap, const float* __restrict__ g_loss, float* __restrict__ d) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	const float e0 = expf(ls[2 * i]), e1 = expf(ls[2 * i + 1]);
	const float m = 0.5f * (e0 + e1), q = gap / m;
	const float sgn = q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f);
	const float c = -sgn * q / m * 0.5f * g_loss[0] / (float)N;          // d|q|/dm = -sgn q / m;  dm/ds_j = e_j / 2
	d[2 * i] = c * e0; d[2 * i + 1] = c * e1;
}

extern "C" int adgs_depth_loss_forward(int n, const float* prediction, const float* target, const float* mask, double* work, float* loss, void* stream_) {
	if (n <= 0) return 0;
	if (!prediction || !target || !work || !loss) { set_error("adgs_depth_loss_forward: NULL pointer"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	const int blocks = std::min((n + 255) / 256, 2048);
	hipLaunchKernelGGL(depth_sums_kernel, dim3(blocks), dim3(256), 0, stream, n, prediction, target, mask, work);
	hipLaunchKernelGGL(depth_solve_kernel, dim3(1), dim3(256), 0, stream, work);
	hipLaunchKernelGGL(depth_l1_kernel, dim3(blocks), dim3(256), 0, stream, n, prediction, target, mask, work);
	hipLaunchKernelGGL(depth_finish_kernel, dim3(1), dim3(256), 0, stream, work, loss);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_depth_loss_backward(int n, const float* prediction, const float* target, const float* mask, const double* work, const float* g_loss,
	float* dL_dprediction, void* stream_) {
	if (n <= 0) return 0;
	if (!prediction || !target || !work || !g_loss || !dL_dprediction) { set_error("adgs_depth_loss_backward: NULL pointer"); return -1; }
	hipLaunchKernelGGL(depth_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, n, prediction, target, mask, work, g_loss, dL_dprediction);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_l1_ssim_forward(int planes, int H, int W, const float* image, const float* gt, double* sums,
	float* d_mu1, float* d_e11, float* d_e12, void* stream) {
	if (planes <= 0 || H <= 0 || W <= 0) return 0;
	if (!image || !gt || !sums) { set_error("adgs_l1_ssim_forward: NULL image / gt / sums"); return -1; }
	if ((d_mu1 != nullptr) != (d_e11 != nullptr) || (d_mu1 != nullptr) != (d_e12 != nullptr)) { set_error("adgs_l1_ssim_forward: pass all three derivative maps or none"); return -1; }
	if (planes > 65535) { set_error("adgs_l1_ssim_forward: more than 65535 planes"); return -1; }
	static const Window win = make_window();
	const dim3 grid((W + TSX - 1) / TSX, (H + TSY - 1) / TSY, planes);
	hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, dim3(LT), 0, (hipStream_t)stream, H, W, image, gt, win, sums, d_mu1, d_e11, d_e12);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

// means of the two sums of adgs_l1_ssim_forward: out2 = (sum |image - gt|, sum ssim_map) / n; the slot rows are consumed (zero afterwards)
namespace {
__global__ void __launch_bounds__(256) l1_ssim_finish_kernel(double* __restrict__ sums, double inv_n, float* __restrict__ out2) {
	__shared__ double s[2][256 / WAVE];
	double a = sums[(size_t)threadIdx.x * 2], b = sums[(size_t)threadIdx.x * 2 + 1];
	sums[(size_t)threadIdx.x * 2] = 0.0; sums[(size_t)threadIdx.x * 2 + 1] = 0.0;
#pragma unroll
	for (int off = WAVE / 2; off > 0; off >>= 1) { a += __shfl_xor(a, off, WAVE); b += __shfl_xor(b, off, WAVE); }
	if ((threadIdx.x & (WAVE - 1)) == 0) { s[0][threadIdx.x / WAVE] = a; s[1][threadIdx.x / WAVE] = b; }
	__syncthreads();
	if (threadIdx.x == 0) {
		double ta = 0, tb = 0;
		for (int w = 0; w < 256 / WAVE; w++) { ta += s[0][w]; tb += s[1][w]; }
		out2[0] = (float)(ta * inv_n); out2[1] = (float)(tb * inv_n);
	}
}
}
extern "C" int adgs_l1_ssim_means(double* sums, long long n, float* out2, void* stream) {
	if (!sums || !out2) { set_error("adgs_l1_ssim_means: NULL pointer"); return -1; }
	static_assert(ADGS_LOSS_SLOTS == 256, "one thread per slot row");
	hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, 1.0 / (double)std::max<long long>(n, 1), out2);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int adgs_l1_ssim_backward(int planes, int H, int W, const float* image, const float* gt,
	const float* d_mu1, const float* d_e11, const float* d_e12, const float* g_l1, const float* g_ssim, float* dL_dimage, void* stream) {
	if (planes <= 0 || H <= 0 || W <= 0) return 0;
	if (!image || !gt || !d_mu1 || !d_e11 || !d_e12 || !dL_dimage) { set_error("adgs_l1_ssim_backward: NULL pointer"); return -1; }
	if (planes > 65535) { set_error("adgs_l1_ssim_backward: more than 65535 planes"); return -1; }
	static const Window win = make_window();
	const dim3 grid((W + TSX - 1) / TSX, (H + TSY - 1) / TSY, planes);
	const float inv_n = (float)(1.0 / ((double)planes * H * W));
	hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(LT), 0, (hipStream_t)stream, H, W, image, gt, win, d_mu1, d_e11, d_e12, g_l1, g_ssim, inv_n, dL_dimage);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}

static FlowCam make_flow_cam(const float* K, const float* R, const float* T, float dist) {
	FlowCam c;
	for (int i = 0; i < 3; i++) {
		for (int j = 0; j < 3; j++) c.M[3 * i + j] = K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j] + K[3 * i + 2] * R[6 + j];
		c.KT[i] = K[3 * i] * T[0] + K[3 * i + 1] * T[1] + K[3 * i + 2] * T[2];
	}
	c.dist = dist;
	return c;
}
static int flow_loss_forward_impl(const char* who, bool dev_cam, int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, double* work, float* loss, void* stream_) {
	if (H <= 0 || W <= 0) return 0;
	if (!img_flow || !flow || !flow_vis || !K || !R || !T || !work || !loss) { set_error(std::string(who) + ": NULL pointer"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	const int n = H * W;
	FlowCam c; FlowCamDev cd{nullptr, nullptr, nullptr};
	if (dev_cam) { c = FlowCam{}; c.dist = dist; cd = FlowCamDev{K, R, T}; } else c = make_flow_cam(K, R, T, dist);
	hipLaunchKernelGGL(flow_loss_sum_kernel, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, stream, H, W, img_flow, flow, flow_vis, img_opacity, c, cd, work);
	hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(256), 0, stream, work, loss);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
static int flow_loss_backward_impl(const char* who, bool dev_cam, int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, const double* work, const float* g_loss, float* dL_dimg_flow, float* dL_dimg_opacity,
	void* stream_) {
	if (H <= 0 || W <= 0) return 0;
	if (!img_flow || !flow || !flow_vis || !K || !R || !T || !work || !g_loss || !dL_dimg_flow) { set_error(std::string(who) + ": NULL pointer"); return -1; }
	const int n = H * W;
	FlowCam c; FlowCamDev cd{nullptr, nullptr, nullptr};
	if (dev_cam) { c = FlowCam{}; c.dist = dist; cd = FlowCamDev{K, R, T}; } else c = make_flow_cam(K, R, T, dist);
	hipLaunchKernelGGL(flow_loss_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, H, W, img_flow, flow, flow_vis, img_opacity,
		c, cd, work, g_loss, dL_dimg_flow, dL_dimg_opacity);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_flow_loss_forward(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, double* work, float* loss, void* stream) {
	return flow_loss_forward_impl("adgs_flow_loss_forward", false, H, W, img_flow, flow, flow_vis, img_opacity, K, R, T, dist, work, loss, stream);
}
extern "C" int adgs_flow_loss_forward_devcam(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, double* work, float* loss, void* stream) {
	return flow_loss_forward_impl("adgs_flow_loss_forward_devcam", true, H, W, img_flow, flow, flow_vis, img_opacity, K, R, T, dist, work, loss, stream);
}
extern "C" int adgs_flow_loss_backward(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, const double* work, const float* g_loss, float* dL_dimg_flow, float* dL_dimg_opacity,
	void* stream) {
	return flow_loss_backward_impl("adgs_flow_loss_backward", false, H, W, img_flow, flow, flow_vis, img_opacity, K, R, T, dist, work, g_loss, dL_dimg_flow,
		dL_dimg_opacity, stream);
}
extern "C" int adgs_flow_loss_backward_devcam(int H, int W, const float* img_flow, const float* flow, const float* flow_vis, const float* img_opacity,
	const float* K, const float* R, const float* T, float dist, const double* work, const float* g_loss, float* dL_dimg_flow, float* dL_dimg_opacity,
	void* stream) {
	return flow_loss_backward_impl("adgs_flow_loss_backward_devcam", true, H, W, img_flow, flow, flow_vis, img_opacity, K, R, T, dist, work, g_loss, dL_dimg_flow,
		dL_dimg_opacity, stream);
}
extern "C" int adgs_bce_clip_forward(int n, const float* pred, const float* target, float lo, float hi, int invert, int positive_target,
	double* work, float* loss, void* stream_) {
	if (n <= 0) return 0;
	if (!pred || !target || !work || !loss) { set_error("adgs_bce_clip_forward: NULL pointer"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	hipLaunchKernelGGL(bce_clip_sum_kernel, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, stream, n, pred, target, lo, hi, invert, positive_target, work);
	hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(256), 0, stream, work, loss);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_bce_clip_backward(int n, const float* pred, const float* target, float lo, float hi, int invert, int positive_target,
	const float* g_loss, float* dL_dpred, void* stream_) {
	if (n <= 0) return 0;
	if (!pred || !target || !g_loss || !dL_dpred) { set_error("adgs_bce_clip_backward: NULL pointer"); return -1; }
	hipLaunchKernelGGL(bce_clip_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, n, pred, target, lo, hi, invert, positive_target, g_loss, dL_dpred);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_group_var_forward(int N, int G, int K, int D, int inner, const float* x, const int64_t* idx, double* work, float* loss, void* stream_) {
	if (G <= 0 || D <= 0) return 0;
	if (!x || !idx || !work || !loss) { set_error("adgs_group_var_forward: NULL pointer"); return -1; }
	if (K < 2 || K > GV_MAXK || inner <= 0 || D % inner != 0 || N <= 0) { set_error("adgs_group_var_forward: needs 2 <= K <= 32 neighbours and D a multiple of inner"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	const long long total = (long long)G * D;
	hipLaunchKernelGGL(group_var_sum_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, stream, N, G, K, D, x,
		reinterpret_cast<const long long*>(idx), (double)G * (double)(D / inner), work);
	hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(256), 0, stream, work, loss);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_group_var_backward(int N, int G, int K, int D, int inner, const float* x, const int64_t* idx, const float* g_loss, float* dL_dx, void* stream_) {
	if (G <= 0 || D <= 0) return 0;
	if (!x || !idx || !g_loss || !dL_dx) { set_error("adgs_group_var_backward: NULL pointer"); return -1; }
	if (K < 2 || K > GV_MAXK || inner <= 0 || D % inner != 0 || N <= 0) { set_error("adgs_group_var_backward: needs 2 <= K <= 32 neighbours and D a multiple of inner"); return -1; }
	const long long total = (long long)G * D;
	const float scale = (float)(2.0 / ((double)(K - 1) * (double)G * (double)(D / inner)));
	hipLaunchKernelGGL(group_var_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, N, G, K, D, x,
		reinterpret_cast<const long long*>(idx), scale, g_loss, dL_dx);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_sigma_loss_forward(int N, const float* log_sigma, float frame_gap, double* work, float* loss, void* stream_) {
	if (N <= 0) return 0;
	if (!log_sigma || !work || !loss) { set_error("adgs_sigma_loss_forward: NULL pointer"); return -1; }
	hipStream_t stream = (hipStream_t)stream_;
	hipLaunchKernelGGL(sigma_loss_sum_kernel, dim3(std::min((N + 255) / 256, 2048)), dim3(256), 0, stream, N, log_sigma, frame_gap, work);
	hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(256), 0, stream, work, loss);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
extern "C" int adgs_sigma_loss_backward(int N, const float* log_sigma, float frame_gap, const float* g_loss, float* dL_dlog_sigma, void* stream_) {
	if (N <= 0) return 0;
	if (!log_sigma || !g_loss || !dL_dlog_sigma) { set_error("adgs_sigma_loss_backward: NULL pointer"); return -1; }
	hipLaunchKernelGGL(sigma_loss_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream_, N, log_sigma, frame_gap, g_loss, dL_dlog_sigma);
	ADGS_HIP_CHECK(hipGetLastError());
	return 0;
}
