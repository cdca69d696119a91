// Probability tables of the PairHMM (host side).  Definitions follow pairhmm/xlnx/host/Context.h:
//   jacobianLogTable   :42-47     matchToMatchProb :50-61 (through approximateLog10SumLog10 :67-90)
//   ph2pr              :105-107 (double, pow) / :145-147 (float, powf)
//   INITIAL_CONSTANT   :109 (2^1020) / :149 (2^120)
// The float tables are built with float intermediates exactly where the reference's Context<float>
// has them, so that the device multiplies by the same bits as compute_full_prob_avxs.
#include <math.h>
#include <mutex>
#include <vector>
#include "accg_internal.h"

namespace accg {
namespace {

constexpr int kMaxQual = 254;
constexpr double kTol = 8.0, kStep = 0.0001;
constexpr int kJac = (int)(kTol / kStep) + 1;

template <typename T>
T log10_sum(const std::vector<T>& jac, T a, T b) {   // log10(10^a + 10^b), table-approximated
  T big = a > b ? a : b, small = a > b ? b : a;
  T diff = big - small;
  if (diff >= (T)kTol) return big;
  T scaled = diff * (T)(1.0 / kStep);
  int idx = scaled > (T)0 ? (int)(scaled + (T)0.5) : (int)(scaled - (T)0.5);
  return big + jac[idx];
}

template <typename T>
void fill_m2m(T* out /*8256 = entries reachable with qualities & 127*/) {
  std::vector<T> jac(kJac);
  for (int k = 0; k < kJac; k++) jac[k] = (T)log10(1.0 + pow(10.0, -((double)k) * kStep));
  const double inv_ln10 = 1.0 / log(10);
  for (int hi = 0, off = 0; hi <= 127; off += ++hi)
    for (int lo = 0; lo <= hi; lo++) {
      double s = log10_sum<T>(jac, (T)(-0.1 * hi), (T)(-0.1 * lo));
      double p = pow(10, s);
      out[off + lo] = (T)pow(10, log1p(-(p < 1.0 ? p : 1.0)) * inv_ln10);
    }
}

HostTables* g_tab = nullptr;
std::once_flag g_once;

void build() {
  HostTables* t = new HostTables;
  static_assert(kMaxQual == 254, "Context.h:8");
  fill_m2m<float>(t->m2m_f);
  fill_m2m<double>(t->m2m_d);
  for (int q = 0; q < 128; q++) {
    t->ph_f[q] = powf(10.f, -((float)q) / 10.f);
    t->ph_d[q] = pow(10.0, -((double)q) / 10.0);
    t->omph_f[q] = 1.0f - t->ph_f[q];          // baseline_impl.cpp:54,81
    t->omph_d[q] = 1.0 - t->ph_d[q];
    t->phd3_f[q] = t->ph_f[q] / 3.0f;          // baseline_impl.cpp:83
    t->phd3_d[q] = t->ph_d[q] / 3.0;
  }
  t->init_f = ldexpf(1.f, 120);
  t->log10_init_f = log10f(t->init_f);
  t->init_d = ldexp(1.0, 1020);
  t->log10_init_d = log10(t->init_d);
  g_tab = t;
}

}  // namespace

const HostTables& host_tables() {
  std::call_once(g_once, build);
  return *g_tab;
}

}  // namespace accg
