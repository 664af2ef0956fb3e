#include "prof.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace serl {
namespace {
struct Pair { hipEvent_t a, b; int key; };
struct State {
  std::mutex mu;
  int period = 0;  // 0 = off, k >= 1: time every k-th launch of each tag
  std::vector<std::string> names;
  std::map<std::string, int> index;
  std::vector<Pair> pairs;       // recorded, not yet resolved
  std::vector<hipEvent_t> pool;  // free events
  std::vector<double> total_ms;
  std::vector<long long> count;
  std::vector<long long> seen;   // launches of the tag since the last reset (timed or not)
};
State g;

hipEvent_t get_event() {
  if (!g.pool.empty()) { hipEvent_t e = g.pool.back(); g.pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void resolve_locked() {
  for (Pair& p : g.pairs) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g.total_ms[p.key] += ms;
      g.count[p.key] += 1;
    }
    g.pool.push_back(p.a);
    g.pool.push_back(p.b);
  }
  g.pairs.clear();
}
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t s) : stream(s) {
  if (!g.period) return;
  std::lock_guard<std::mutex> l(g.mu);
  auto it = g.index.find(name);
  int key;
  if (it == g.index.end()) {
    key = (int)g.names.size();
    g.names.push_back(name);
    g.index[name] = key;
    g.total_ms.push_back(0.0);
    g.count.push_back(0);
    g.seen.push_back(0);
  } else key = it->second;
  // sampled (the events themselves cost a few us on the stream); the phase drifts by one every `period`
  // launches so that a tag launched `period` times per step with different shapes is sampled evenly
  const long long n = g.seen[key]++;
  if ((n + n / g.period) % g.period) return;
  Pair p{get_event(), get_event(), key};
  (void)hipEventRecord(p.a, stream);
  g.pairs.push_back(p);
  slot = (int)g.pairs.size() - 1;
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> l(g.mu);
  if (slot < (int)g.pairs.size()) (void)hipEventRecord(g.pairs[slot].b, stream);
}
}  // namespace serl

extern "C" {
int serl_profile_enable(int on) {
  std::lock_guard<std::mutex> l(serl::g.mu);
  serl::g.period = on > 0 ? on : 0;
  return SERL_OK;
}
int serl_profile_reset(void) {
  std::lock_guard<std::mutex> l(serl::g.mu);
  serl::resolve_locked();
  for (auto& v : serl::g.total_ms) v = 0.0;
  for (auto& v : serl::g.count) v = 0;
  for (auto& v : serl::g.seen) v = 0;
  return SERL_OK;
}
int serl_profile_read(int max_entries, char* names /* [max][64] */, double* total_ms, int64_t* counts, int* n_out) {
  SERL_REQUIRE(names && total_ms && counts && n_out, "NULL argument");
  std::lock_guard<std::mutex> l(serl::g.mu);
  serl::resolve_locked();
  int n = 0;
  for (size_t i = 0; i < serl::g.names.size() && n < max_entries; ++i) {
    if (serl::g.count[i] == 0) continue;
    snprintf(names + (size_t)n * 64, 64, "%s", serl::g.names[i].c_str());
    total_ms[n] = serl::g.total_ms[i];
    counts[n] = serl::g.count[i];
    ++n;
  }
  *n_out = n;
  return SERL_OK;
}
}
