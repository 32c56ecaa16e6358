/* Plain C99 consumer of include/apdgicp_hip.h: proves that the boundary is a C ABI (no C++ in the header) and that the
 * library links and fails loudly -- with a message, not a crash -- when no GPU is present.
 * With a GPU (argv[1] == "gpu") it registers two tiny clouds through the batch API. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "apdgicp_hip.h"

int main(int argc, char** argv) {
  apdgicp_params p;
  apdgicp_default_params(&p);
  if (p.k_correspondences != 20 || p.max_iterations != 64) return 10;
  printf("abi %d k %d\n", apdgicp_abi_version(), p.k_correspondences);
  apdgicp_handle* h = NULL;
  int rc = apdgicp_create(&p, 0, NULL, &h);
  if (argc < 2 || strcmp(argv[1], "gpu") != 0) {
    /* no GPU expected: the call must fail with a status and an error text */
    if (rc == 0) {
      apdgicp_destroy(h);
      printf("gpu present\n");
      return 0;
    }
    printf("create failed as expected: %d (%s)\n", rc, apdgicp_last_error());
    return (rc < 0 && apdgicp_last_error()[0] != 0) ? 0 : 11;
  }
  if (rc != 0) return 12;
  enum { N = 256 };
  float* src = (float*)malloc(sizeof(float) * 3 * N);
  float* tgt = (float*)malloc(sizeof(float) * 3 * N);
  unsigned s = 12345u;
  for (int i = 0; i < 3 * N; i++) {
    s = s * 1664525u + 1013904223u;
    tgt[i] = (float)(s >> 8) / 16777216.0f * 20.0f;
    src[i] = tgt[i] + (i % 3 == 0 ? 0.05f : 0.0f); /* the source is the target shifted by 5 cm in x */
  }
  if (apdgicp_set_target(h, tgt, N, 12, 0, 1) != 0 || apdgicp_set_source(h, src, N, 12, 0, 2) != 0) return 13;
  apdgicp_result r;
  if (apdgicp_align(h, NULL, &r) != 0) return 14;
  printf("converged %d iterations %d tx %.4f\n", r.converged, r.iterations, r.T[12]);
  apdgicp_destroy(h);
  free(src);
  free(tgt);
  return (r.converged && r.T[12] < -0.03f && r.T[12] > -0.07f) ? 0 : 15;
}
