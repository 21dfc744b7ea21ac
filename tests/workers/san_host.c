/* Host-side sweep of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizer build of the host shim).
 * Only entry points that never touch a device are called with valid arguments: version / error / counters, tuning, every
 * workspace and algorithm query over the layer geometries of the models plus degenerate and hostile descriptors; the compute
 * entry points are called ONLY with arguments they must reject before any launch (null tensors, invalid descriptors).
 * Built and run by tools/sanitize_host.sh against lib/san/libt2i_hip_san.so; a sanitizer report aborts the run (non-zero exit). */
#include <limits.h>
#include <stdio.h>
#include <string.h>

#include "t2i_hip.h"

static t2i_conv_desc desc(int B, int H, int W, int Cin, int Cout, int K, int S, int math) {
  t2i_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.KH = d.KW = K; d.SH = d.SW = S;
  d.Ho = (H + S - 1) / S; d.Wo = (W + S - 1) / S;            /* TF SAME */
  { int ph = (d.Ho - 1) * S + K - H, pw = (d.Wo - 1) * S + K - W; d.pad_t = ph > 0 ? ph / 2 : 0; d.pad_l = pw > 0 ? pw / 2 : 0; }
  d.math = math;
  return d;
}

int main(void) {
  unsigned long long acc = 0;
  int calls = 0, rejected = 0;
  if (t2i_version() < 8) { fprintf(stderr, "unexpected ABI version %d\n", t2i_version()); return 2; }
  if (!t2i_last_error()) return 2;
  if (t2i_stat("no_such_counter") != -1) return 2;
  acc += (unsigned long long)t2i_stat("pair_fused");
  /* tuning: a known key round-trips, an unknown key is refused, a NULL key is refused */
  if (t2i_tuning_set("winograd", 1.0) != T2I_OK) return 2;
  if (t2i_tuning_set("no_such_key", 1.0) == T2I_OK) return 2;
  if (t2i_tuning_set(NULL, 1.0) == T2I_OK) return 2;
  /* geometry sweep: the wgancls / gancls / StackGAN / PGGAN layer classes at the batch sizes the tests and the bench use */
  static const int shapes[][6] = {   /* H, Cin, Cout, K, S, (unused) */
      {64, 3, 128, 4, 2, 0}, {32, 128, 256, 4, 2, 0}, {16, 256, 512, 4, 2, 0}, {8, 512, 1024, 4, 2, 0}, {4, 1024, 256, 1, 1, 0},
      {4, 256, 512, 3, 1, 0}, {4, 512, 1024, 3, 1, 0}, {4, 1152, 1024, 3, 1, 0}, {4, 1024, 1024, 1, 1, 0}, {4, 1024, 1, 4, 4, 0},
      {1, 256, 16384, 1, 1, 0}, {8, 512, 512, 3, 1, 0}, {8, 128, 512, 3, 1, 0}, {16, 256, 256, 3, 1, 0}, {32, 128, 128, 3, 1, 0},
      {64, 3, 3, 3, 1, 0}, {64, 64, 64, 3, 1, 0}, {128, 32, 32, 3, 1, 0}, {256, 16, 3, 3, 1, 0}, {4, 640, 512, 1, 1, 0}, {5, 7, 9, 3, 1, 0},
      {6, 130, 34, 3, 1, 0}, {10, 36, 20, 4, 2, 0}};
  static const int batches[] = {1, 3, 8, 24, 64, 192};
  for (size_t s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s)
    for (size_t b = 0; b < sizeof(batches) / sizeof(batches[0]); ++b)
      for (int math = 0; math < 2; ++math) {
        t2i_conv_desc d = desc(batches[b], shapes[s][0], shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], math);
        acc += t2i_conv2d_workspace_bytes(&d); acc += t2i_conv2d_input_transform_bytes(&d); acc += t2i_conv2d_stats_bytes(&d);
        for (int which = 0; which < 3; ++which) acc += (unsigned long long)(t2i_conv2d_algo(&d, which) + 1);
        calls += 6;
      }
  /* hostile descriptors: zero / negative / overflowing dimensions must be refused (algo -1, workspace 0 or an error), not crash */
  {
    static const int bad[][5] = {{0, 4, 4, 4, 4}, {-1, 4, 4, 4, 4}, {1, 0, 4, 4, 4}, {1, 4, -4, 4, 4}, {1, 4, 4, 0, 4}, {1, 4, 4, 4, 0},
                                 {INT_MAX, INT_MAX, INT_MAX, 4, 4}, {INT_MAX, 4, 4, INT_MAX, INT_MAX}, {1, 46341, 46341, 1024, 1024}};
    for (size_t i = 0; i < sizeof(bad) / sizeof(bad[0]); ++i) {
      t2i_conv_desc d = desc(1, 4, 4, 4, 4, 3, 1, 0);
      d.B = bad[i][0]; d.H = bad[i][1]; d.W = bad[i][2]; d.Cin = bad[i][3]; d.Cout = bad[i][4];
      acc += t2i_conv2d_workspace_bytes(&d); acc += t2i_conv2d_input_transform_bytes(&d); acc += t2i_conv2d_stats_bytes(&d);
      for (int which = 0; which < 4; ++which) acc += (unsigned long long)(t2i_conv2d_algo(&d, which) + 1);
      d = desc(2, 8, 8, 16, 16, 3, 1, 0);
      d.KH = bad[i][0]; d.SH = bad[i][1] % 7; d.pad_t = bad[i][2];
      acc += t2i_conv2d_workspace_bytes(&d);
      acc += (unsigned long long)(t2i_conv2d_algo(&d, 0) + 1);
      calls += 10;
    }
    acc += t2i_conv2d_workspace_bytes(NULL);
    acc += (unsigned long long)(t2i_conv2d_algo(NULL, 0) + 1);
  }
  /* the other workspace queries */
  {
    static const long long rows[] = {0, 1, 63, 64, 1024, 196608, 12582912, (1LL << 40)};
    static const int C[] = {1, 3, 4, 64, 128, 1024, 16384};
    for (size_t r = 0; r < sizeof(rows) / sizeof(rows[0]); ++r)
      for (size_t c = 0; c < sizeof(C) / sizeof(C[0]); ++c) {
        acc += t2i_col_reduce_workspace_bytes(rows[r], C[c]); acc += t2i_bn_bwd_fused_workspace_bytes(rows[r], C[c]);
        for (int g = 1; g <= 3; ++g) acc += t2i_bn_grouped_workspace_bytes(rows[r], C[c], g);
        calls += 5;
      }
    for (int B = 0; B <= 192; B += 8) { acc += t2i_row_moments_workspace_bytes(B); ++calls; }
  }
  /* compute entry points with arguments they must reject BEFORE any launch */
  {
    t2i_conv_desc d = desc(8, 4, 4, 256, 512, 3, 1, 0);
    float dummy[4] = {0};
    rejected += t2i_conv2d_fwd(&d, NULL, NULL, NULL, NULL, T2I_ACT_NONE, 0.f, NULL, NULL, 0, NULL) != T2I_OK;
    rejected += t2i_conv2d_bwd_data(&d, NULL, NULL, NULL, NULL, T2I_ACT_NONE, 0.f, NULL, NULL, 0, NULL) != T2I_OK;
    rejected += t2i_conv2d_bwd_filter(&d, NULL, NULL, NULL, 0, NULL, NULL, 0, NULL) != T2I_OK;
    d.B = 0;
    rejected += t2i_conv2d_fwd(&d, dummy, dummy, NULL, dummy, T2I_ACT_NONE, 0.f, NULL, NULL, 0, NULL) != T2I_OK;
    rejected += t2i_conv2d_bwd_data(&d, dummy, dummy, NULL, dummy, T2I_ACT_NONE, 0.f, NULL, NULL, 0, NULL) != T2I_OK;
    rejected += t2i_conv2d_bwd_filter(&d, dummy, dummy, dummy, 0, NULL, NULL, 0, NULL) != T2I_OK;
    /* ABI v9: xform_valid_rows / xform_plane_rows outside their range, the new small entry points with bad arguments */
    d = desc(64, 8, 8, 512, 512, 3, 1, 0);                     /* forward conv and filter gradient share a Winograd transform here */
    t2i_conv_opts o;
    memset(&o, 0, sizeof(o));
    _Alignas(16) static float big[64];
    o.xform = big; o.xform_bytes = (size_t)1 << 40; o.xform_mode = T2I_XFORM_HAVE; o.xform_valid_rows = -1;
    rejected += t2i_conv2d_bwd_filter(&d, dummy, dummy, dummy, 0, &o, NULL, 0, NULL) != T2I_OK && strstr(t2i_last_error(), "xform_valid_rows") != NULL;
    o.xform_valid_rows = 0; o.xform_plane_rows = 4;            /* fewer images than the batch */
    rejected += t2i_conv2d_bwd_filter(&d, dummy, dummy, dummy, 0, &o, NULL, 0, NULL) != T2I_OK && strstr(t2i_last_error(), "xform_plane_rows") != NULL;
    rejected += t2i_trunc_normal(NULL, 16, 1, 0, 0.f, 1.f, -2.f, 2.f, NULL) != T2I_OK;
    rejected += t2i_trunc_normal(dummy, 4, 1, 0, 0.f, 0.f, -2.f, 2.f, NULL) != T2I_OK;
    rejected += t2i_trunc_normal(dummy, 4, 1, 0, 0.f, 1.f, 2.f, -2.f, NULL) != T2I_OK;
    rejected += t2i_zero_ranges(dummy, NULL, 3, NULL) != T2I_OK;
    rejected += t2i_zero_ranges(dummy, (const int64_t*)big, 0, NULL) != T2I_OK;
    if (rejected != 13) { fprintf(stderr, "only %d of 13 invalid calls were rejected\n", rejected); return 3; }
    if (!t2i_last_error()[0]) { fprintf(stderr, "no error message after a rejected call\n"); return 3; }
  }
  printf("san_host: %d host-side queries, %d invalid compute calls rejected, no sanitizer report (checksum %llu)\n", calls, rejected, acc);
  return 0;
}
