/* A plain-C (gcc -std=c99) consumer of the batched C-ABI, the engine and the multi-GPU helpers: the headers are valid C, the libraries link from
 * C, and a host with nothing but a C FFI can drive the hot path.  Frame preparation -> pyramid -> Sobel -> fused Gauss-Newton evaluation of two
 * lanes with an identity transform (current frame == keyframe): b must vanish against A (analytic known answer), lanes must be independent;
 * then one engine step pair and the chunk partition / composition helpers. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rgbid_batched.h"
#include "rgbid_dist.h"

#define CHECK(x) do { int e_ = (x); if (e_ != 0) { printf("FAILED %s -> %d (%s)\n", #x, e_, rgbid_error_string(e_)); return 1; } else printf("ok %s\n", #x); } while (0)

enum { ROWS = 48, COLS = 64, LANES = 2 };

static rgbid_imgb alloc_maps(size_t elem, int rows, void** keep) {
  rgbid_imgb m;
  size_t step = 0;
  void* p = NULL;
  rgbid_malloc_pitch(&p, &step, (size_t)COLS * elem, (size_t)rows * LANES);
  m.data = p; m.step = step; m.lane_stride = step * rows; m.rows = rows; m.cols = COLS;
  *keep = p;
  return m;
}

int main(void) {
  rgbid_ctx* ctx = NULL;
  CHECK(rgbid_ctx_create(&ctx, 0, NULL));
  /* synthetic frames: a tilted plane + texture, lane 1 a different plane */
  static unsigned short depth[LANES][ROWS][COLS];
  static unsigned char rgb[LANES][ROWS][COLS][3];
  for (int l = 0; l < LANES; ++l)
    for (int y = 0; y < ROWS; ++y)
      for (int x = 0; x < COLS; ++x) {
        depth[l][y][x] = (unsigned short)(1500 + 200 * l + 3 * x + 2 * y);
        unsigned char v = (unsigned char)(100 + 50 * sin(0.4 * x + 0.2 * l) * cos(0.3 * y) + ((x * 7 + y * 13) % 11));
        rgb[l][y][x][0] = v; rgb[l][y][x][1] = (unsigned char)(v / 2 + 40); rgb[l][y][x][2] = (unsigned char)(255 - v);
      }
  depth[0][5][7] = 0; /* an invalid pixel */
  void *kd, *kc, *k[12];
  rgbid_imgb d16 = alloc_maps(2, ROWS, &kd), c8 = alloc_maps(3, ROWS, &kc);
  rgbid_imgb iD = alloc_maps(4, ROWS, &k[0]), I = alloc_maps(4, ROWS, &k[1]), r = alloc_maps(4, ROWS, &k[2]), g = alloc_maps(4, ROWS, &k[3]), b_ = alloc_maps(4, ROWS, &k[4]);
  rgbid_imgb gWx = alloc_maps(4, ROWS, &k[5]), gWy = alloc_maps(4, ROWS, &k[6]), gIx = alloc_maps(4, ROWS, &k[7]), gIy = alloc_maps(4, ROWS, &k[8]);
  for (int l = 0; l < LANES; ++l) {
    CHECK(rgbid_memcpy2d_h2d(ctx, (char*)d16.data + l * d16.lane_stride, d16.step, depth[l], COLS * 2, COLS * 2, ROWS));
    CHECK(rgbid_memcpy2d_h2d(ctx, (char*)c8.data + l * c8.lane_stride, c8.step, rgb[l], COLS * 3, COLS * 3, ROWS));
  }
  CHECK(rgbid_prep_frame_batched(ctx, LANES, &d16, &c8, &iD, &I, &r, &g, &b_, 1.0f, NULL));
  CHECK(rgbid_compute_gradient_batched(ctx, LANES, &iD, &gWx, &gWy, NULL));
  CHECK(rgbid_compute_gradient_batched(ctx, LANES, &I, &gIx, &gIy, NULL));
  float w00 = 0.f;
  CHECK(rgbid_memcpy_d2h(ctx, &w00, iD.data, 4));
  if (fabsf(w00 - 1000.f / 1500.f) > 1e-6f) { printf("FAILED inverse depth %g\n", w00); return 1; }
  /* identity transform in projected form: K I K^-1 = I, K 0 = 0 */
  float R[LANES][9] = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {1, 0, 0, 0, 1, 0, 0, 0, 1}}, t[LANES][3] = {{0, 0, 0}, {0, 0, 0}};
  rgbid_intr K = {52.5f, 52.5f, 31.5f, 23.5f};
  rgbid_sys_params sp[LANES];
  for (int l = 0; l < LANES; ++l) {
    sp[l].sigma_depthinv = 0.0025f; sp[l].sigma_int = 5.f; sp[l].bias_depthinv = 0.f; sp[l].bias_int = 0.f; sp[l].nu_depthinv = 5.f; sp[l].nu_int = 5.f;
    sp[l].mestimator = RGBID_STUDENT; sp[l].weighting = RGBID_INDEPENDENT; sp[l].student_nu = 1; sp[l].nu_int_from_max = 0;
  }
  double A[LANES][36], bb[LANES][6];
  float ms = 0.f;
  for (int numerics = 0; numerics < 2; ++numerics) {
    CHECK(rgbid_gn_fused_batched(ctx, LANES, &iD, &I, &gWx, &gWy, &gIx, &gIy, &iD, &I, &R[0][0], &t[0][0], K, sp, numerics, RGBID_WM_AUTO, &A[0][0], &bb[0][0], &ms));
    for (int l = 0; l < LANES; ++l) {
      double amax = 0, bmax = 0;
      for (int i = 0; i < 36; ++i) if (fabs(A[l][i]) > amax) amax = fabs(A[l][i]);
      for (int i = 0; i < 6; ++i) if (fabs(bb[l][i]) > bmax) bmax = fabs(bb[l][i]);
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) if (A[l][i * 6 + j] != A[l][j * 6 + i]) { printf("FAILED symmetry\n"); return 1; }
      if (!(amax > 0) || !(A[l][0] > 0) || bmax > 1e-6 * amax) { printf("FAILED identity KAT lane %d: |A| %g |b| %g\n", l, amax, bmax); return 1; }
    }
    if (memcmp(A[0], A[1], sizeof(A[0])) == 0) { printf("FAILED lanes are not independent\n"); return 1; }
    printf("ok identity known answer, numerics %d (%.3f ms)\n", numerics, ms);
  }
  /* a FAST call on a geometry the fast kernels cannot take is an error, not another class */
  { rgbid_imgb odd = iD; odd.cols = COLS - 1;
    rgbid_imgb o2 = I, o3 = gWx, o4 = gWy, o5 = gIx, o6 = gIy; o2.cols = o3.cols = o4.cols = o5.cols = o6.cols = COLS - 1;
    int e = rgbid_gn_fused_batched(ctx, LANES, &odd, &o2, &o3, &o4, &o5, &o6, &odd, &o2, &R[0][0], &t[0][0], K, sp, RGBID_NUMERICS_FAST, RGBID_WM_AUTO, &A[0][0], &bb[0][0], NULL);
    if (e != RGBID_E_INVALID) { printf("FAILED fast on odd geometry returned %d\n", e); return 1; }
    printf("ok fast numerics refused on 63 columns\n"); }
  /* the engine from C: two frames of the same scene -> a tracked frame with a near-identity pose */
  rgbid_engine_config cfg;
  rgbid_engine_default_config(&cfg);
  cfg.rows = ROWS; cfg.cols = COLS; cfg.lanes = LANES; cfg.levels = 2; cfg.iters[0] = 4; cfg.iters[1] = 2; cfg.iters[2] = 0;
  cfg.fx = K.fx; cfg.fy = K.fy; cfg.cx = K.cx; cfg.cy = K.cy; cfg.use_graph = 0; cfg.record_capacity = 4;
  rgbid_engine* eng = NULL;
  CHECK(rgbid_engine_create(&eng, ctx, &cfg));
  void *dd = NULL, *dc = NULL;
  CHECK(rgbid_malloc(&dd, sizeof(depth))); CHECK(rgbid_malloc(&dc, sizeof(rgb)));
  CHECK(rgbid_memcpy_h2d(ctx, dd, depth, sizeof(depth))); CHECK(rgbid_memcpy_h2d(ctx, dc, rgb, sizeof(rgb)));
  CHECK(rgbid_engine_step(eng, dd, dc));
  /* the second frame: the same view with sensor noise (identical frames have zero residuals, and the estimated scale sigma = 0 makes the reference
   * algorithm itself divide by zero) */
  for (int l = 0; l < LANES; ++l)
    for (int y = 0; y < ROWS; ++y)
      for (int x = 0; x < COLS; ++x) {
        if (depth[l][y][x]) depth[l][y][x] = (unsigned short)(depth[l][y][x] + ((x * 5 + y * 3 + l) % 5) - 2);
        for (int c = 0; c < 3; ++c) { int v = rgb[l][y][x][c] + ((x * 3 + y * 7 + c) % 5) - 2; rgb[l][y][x][c] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
      }
  void *dd2 = NULL, *dc2 = NULL;
  CHECK(rgbid_malloc(&dd2, sizeof(depth))); CHECK(rgbid_malloc(&dc2, sizeof(rgb)));
  CHECK(rgbid_memcpy_h2d(ctx, dd2, depth, sizeof(depth))); CHECK(rgbid_memcpy_h2d(ctx, dc2, rgb, sizeof(rgb)));
  CHECK(rgbid_engine_step(eng, dd2, dc2));
  rgbid_pose_record rec[2 * LANES];
  CHECK(rgbid_engine_read_records(eng, 0, 2, rec));
  for (int l = 0; l < LANES; ++l) {
    const rgbid_pose_record* p = &rec[LANES + l];
    double tn = sqrt(p->t[0] * p->t[0] + p->t[1] * p->t[1] + p->t[2] * p->t[2]);
    if (!(p->status & RGBID_ST_TRACKED) || tn > 2e-3 || fabs(p->R[0] - 1) > 1e-5) { printf("FAILED engine lane %d status %d |t| %g\n", l, p->status, tn); return 1; }
  }
  printf("ok engine tracked the noisy repeat of the frame at (near) identity\n");
  CHECK(rgbid_engine_destroy(eng));
  /* partition + composition helpers (host arithmetic) */
  int first[3], last[3], start, count;
  CHECK(rgbid_dist_chunk_ranges(10, 3, first, last));
  CHECK(rgbid_dist_rank_chunks(3, 2, 1, &start, &count));
  if (first[0] != 0 || last[2] != 9 || last[0] != first[1] || start != 2 || count != 1) { printf("FAILED partition\n"); return 1; }
  rgbid_free(dd); rgbid_free(dc); rgbid_free(dd2); rgbid_free(dc2); rgbid_free(kd); rgbid_free(kc);
  for (int i = 0; i < 9; ++i) rgbid_free(k[i]);
  CHECK(rgbid_ctx_destroy(ctx));
  printf("all ok\n");
  return 0;
}
