// Replay of the reference's integration harness (pairhmm/host/main.cpp:226-425) on libaccg_compat.so:
//   host_tb <dir> <first> <last>     reads <dir>/input<i>, <dir>/output<i>
// For each file: get_input -> serialize -> compute_fpga -> the post-process of PairHMMWorker::getOutput
// (client/PairHMMWorker.cpp:176-190, fp64 values through compute_fp_avxd) -> compare with the golden doubles under
// the reference's own rule |(got - golden) / golden| <= 5e-3 and not NaN (main.cpp:380-384).  Exit 1 if any batch fails.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../acc_genomics_amd/csrc/compat/accg_compat.h"

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s dir first last\n", argv[0]); return 2; }
  int failed = 0, ran = 0;
  double cells_total = 0;
  for (int i = atoi(argv[2]); i <= atoi(argv[3]); i++) {
    std::string fin = std::string(argv[1]) + "/input" + std::to_string(i), fout = std::string(argv[1]) + "/output" + std::to_string(i);
    int nr = 0, nh = 0; read_t* reads = NULL; hap_t* haps = NULL;
    get_input(nr, nh, reads, haps, fin.c_str());
    std::vector<double> golden((size_t)nr * nh);
    if (get_output(golden.data(), nr * nh, fout.c_str())) return 2;
    uint64_t cells = 0;
    for (int a = 0; a < nr; a++) for (int b = 0; b < nh; b++) cells += (uint64_t)reads[a].len * haps[b].len;
    std::string rs = serialize(reads, nr), hs = serialize(haps, nh);
    float* raw = compute_fpga("unused", rs, hs, cells);
    if (!raw) { printf("batch %d: Skipped\n", i); free_reads(reads, nr); free_haps(haps, nh); continue; }
    const double l10f = log10f(ldexpf(1.f, 120)), l10d = log10(ldexp(1.0, 1020));
    int bad = 0;
    for (int a = 0; a < nr; a++)
      for (int b = 0; b < nh; b++) {
        float f = raw[(size_t)a * nh + b];
        double got;
        if (f < 1e-28f) {
          testcase tc = {reads[a].len, haps[b].len, reads[a]._q, reads[a]._i, reads[a]._d, reads[a]._c, haps[b]._b, reads[a]._b};
          got = log10(compute_fp_avxd(&tc)) - l10d;
        } else got = (double)(log10f(f) - (float)l10f);
        double g = golden[(size_t)a * nh + b];
        if (!(fabs((got - g) / g) <= 5e-3) || got != got) { if (bad < 3) printf("batch %d pair %d,%d: %.9g vs golden %.9g\n", i, a, b, got, g); bad++; }
      }
    printf("batch %d: %d reads x %d haps, %s, kernel %.2f GCUPS\n", i, nr, nh, bad ? "FAILED" : "ok", curr_kernel_gcups);
    failed += bad != 0; ran++; cells_total += (double)cells;
    free_reads(reads, nr); free_haps(haps, nh);
  }
  printf("host_tb: %d batches, %d failed, peak kernel %.1f GCUPS\n", ran, failed, peak_kernel_gcups);
  cleanup();
  return failed ? 1 : 0;
}
