// Action codec on the device (gfx950): the integer work between the environment's factored actions and the policy's
// joint action indices, so that BC labels and IDM / policy outputs never leave HBM.  One thread per action row;
// everything is a few bytes per row, i.e. launch- and HBM-latency-bound -- no tiling to speak of.
//
//  vpt_camera_discretize_kernel   : CameraQuantizer.discretize  (lib/actions.py:88-98): clip, optional mu-law
//        companding, round-half-to-even onto the bin grid (np.round semantics = rint), computed in fp64.
//  vpt_camera_undiscretize_kernel : CameraQuantizer.undiscretize (lib/actions.py:100-108), fp64.
//  vpt_action_from_factored_kernel: CameraHierarchicalMapping.from_factored (lib/action_mapping.py:179-207) incl.
//        factored_buttons_to_groups (:66-104): per mutually exclusive group the LAST pressed button wins, except
//        forward+back and left+right which cancel; inventory (== 1) overrides everything and nulls the camera; the
//        joint index is the mixed-radix number over (hotbar 10, fore_back 3, left_right 3, sprint_sneak 3, use 2,
//        drop 2, attack 2, jump 2, camera 2) in itertools.product order, 8640 = inventory.
//  vpt_action_to_factored_kernel  : CameraHierarchicalMapping.to_factored (:209-219) without the lookup tables: the
//        digits of the joint index are decoded directly; the camera is nulled when the camera meta action is off
//        (not for inventory: the reference's table leaves that flag False).
// Button order = Buttons.ALL (lib/actions.py:21-33): attack, back, forward, jump, left, right, sneak, sprint, use,
// drop, inventory, hotbar.1 .. hotbar.9.
#include "vpt_common.h"
#include "vpt_kernels.h"

#define B_ATTACK 0
#define B_BACK 1
#define B_FORWARD 2
#define B_JUMP 3
#define B_LEFT 4
#define B_RIGHT 5
#define B_SNEAK 6
#define B_SPRINT 7
#define B_USE 8
#define B_DROP 9
#define B_INVENTORY 10
#define B_HOTBAR1 11
#define N_BUTTONS 20
#define JOINT_INVENTORY 8640

__global__ __launch_bounds__(256) void vpt_camera_discretize_kernel(const double* __restrict__ xy, long* __restrict__ out, long n,
                                                                    double maxval, double binsize, double mu, int mu_law) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = fmin(fmax(xy[i], -maxval), maxval);
  if (mu_law) {
    v = v / maxval;
    const double s = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
    v = s * (log(1.0 + mu * fabs(v)) / log(1.0 + mu));
    v *= maxval;
  }
  out[i] = (long)rint((v + maxval) / binsize);
}

__global__ __launch_bounds__(256) void vpt_camera_undiscretize_kernel(const long* __restrict__ pq, double* __restrict__ out, long n,
                                                                      double maxval, double binsize, double mu, int mu_law) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = (double)pq[i] * binsize - maxval;
  if (mu_law) {
    v = v / maxval;
    const double s = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
    v = s * (1.0 / mu) * (pow(1.0 + mu, fabs(v)) - 1.0);
    v *= maxval;
  }
  out[i] = v;
}

__global__ __launch_bounds__(256) void vpt_action_from_factored_kernel(const long* __restrict__ buttons, const long* __restrict__ camera,
                                                                       long* __restrict__ joint_buttons, long* __restrict__ joint_camera,
                                                                       long n, int n_camera_bins) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long* b = buttons + i * N_BUTTONS;
  const int null_bin = n_camera_bins / 2;
  int hotbar = 0;
#pragma unroll
  for (int k = 1; k <= 9; ++k)
    if (b[B_HOTBAR1 + k - 1] != 0) hotbar = k;                       // later button wins
  const bool fwd = b[B_FORWARD] != 0, back = b[B_BACK] != 0;
  const int fore_back = (fwd && back) ? 0 : (back ? 2 : (fwd ? 1 : 0)); // both pressed cancel
  const bool left = b[B_LEFT] != 0, right = b[B_RIGHT] != 0;
  const int left_right = (left && right) ? 0 : (right ? 2 : (left ? 1 : 0));
  const int sprint_sneak = (b[B_SNEAK] != 0) ? 2 : ((b[B_SPRINT] != 0) ? 1 : 0);   // sneak is later in the group
  const long c0 = camera[2 * i], c1 = camera[2 * i + 1];
  const int camera_on = !(c0 == null_bin && c1 == null_bin);
  long jb = hotbar;
  jb = jb * 3 + fore_back;
  jb = jb * 3 + left_right;
  jb = jb * 3 + sprint_sneak;
  jb = jb * 2 + (b[B_USE] != 0);
  jb = jb * 2 + (b[B_DROP] != 0);
  jb = jb * 2 + (b[B_ATTACK] != 0);
  jb = jb * 2 + (b[B_JUMP] != 0);
  jb = jb * 2 + camera_on;
  long jc = c0 * n_camera_bins + c1;
  if (b[B_INVENTORY] == 1) {                                           // exclusive with everything, camera included
    jb = JOINT_INVENTORY;
    jc = (long)null_bin * n_camera_bins + null_bin;
  }
  joint_buttons[i] = jb;
  joint_camera[i] = jc;
}

__global__ __launch_bounds__(256) void vpt_action_to_factored_kernel(const long* __restrict__ joint_buttons, const long* __restrict__ joint_camera,
                                                                     long* __restrict__ buttons, long* __restrict__ camera, long n, int n_camera_bins) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long* b = buttons + i * N_BUTTONS;
#pragma unroll
  for (int k = 0; k < N_BUTTONS; ++k) b[k] = 0;
  const int null_bin = n_camera_bins / 2;
  long jb = joint_buttons[i];
  const long jc = joint_camera[i];
  bool camera_off = false;
  if (jb == JOINT_INVENTORY) {
    b[B_INVENTORY] = 1;
  } else {
    camera_off = (jb % 2) == 0; jb /= 2;
    if (jb % 2) b[B_JUMP] = 1;
    jb /= 2;
    if (jb % 2) b[B_ATTACK] = 1;
    jb /= 2;
    if (jb % 2) b[B_DROP] = 1;
    jb /= 2;
    if (jb % 2) b[B_USE] = 1;
    jb /= 2;
    const int ss = (int)(jb % 3); jb /= 3;
    if (ss == 1) b[B_SPRINT] = 1; else if (ss == 2) b[B_SNEAK] = 1;
    const int lr = (int)(jb % 3); jb /= 3;
    if (lr == 1) b[B_LEFT] = 1; else if (lr == 2) b[B_RIGHT] = 1;
    const int fb = (int)(jb % 3); jb /= 3;
    if (fb == 1) b[B_FORWARD] = 1; else if (fb == 2) b[B_BACK] = 1;
    if (jb > 0) b[B_HOTBAR1 + jb - 1] = 1;
  }
  camera[2 * i] = camera_off ? null_bin : jc / n_camera_bins;
  camera[2 * i + 1] = camera_off ? null_bin : jc % n_camera_bins;
}

static inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

extern "C" int vpt_camera_codec_launch(int decode, const void* in, void* out, long n, double maxval, double binsize, double mu, int mu_law,
                                       hipStream_t stream) {
  if (n <= 0 || maxval <= 0.0 || binsize <= 0.0 || (mu_law && mu <= 0.0)) return -1;
  if (!decode) hipLaunchKernelGGL(vpt_camera_discretize_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, (const double*)in, (long*)out, n, maxval, binsize, mu, mu_law);
  else hipLaunchKernelGGL(vpt_camera_undiscretize_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, (const long*)in, (double*)out, n, maxval, binsize, mu, mu_law);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int vpt_action_mapping_launch(int to_factored, const long* a, const long* b, long* oa, long* ob, long n, int n_camera_bins, hipStream_t stream) {
  if (n <= 0 || n_camera_bins < 1 || !(n_camera_bins & 1)) return -1;
  if (!to_factored) hipLaunchKernelGGL(vpt_action_from_factored_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, a, b, oa, ob, n, n_camera_bins);
  else hipLaunchKernelGGL(vpt_action_to_factored_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, a, b, oa, ob, n, n_camera_bins);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
