/* Test harness (tests/ only): the oracle's own statement of libstdc++'s introselect (oracle/evz_orb.cpp) against the real
 * std::nth_element of the compiler in this image, on random sequences with few distinct keys (ties decide everything) and on
 * sequences built to exhaust the depth limit.  Prints "mismatches M of T, depth-limit cases D"; exit code 0 iff M == 0. */
#include "../../oracle/evz_orb.cpp"
#include <random>
int main() {
  std::mt19937 rng(5);
  long bad = 0, tot = 0, deep = 0;
  auto comp = [](const std::pair<float, int>& x, const std::pair<float, int>& y) { return x.first > y.first; };
  auto run = [&](std::vector<std::pair<float, int>> a, int nth) {
    auto b = a;
    std::nth_element(a.begin(), a.begin() + nth, a.end(), comp);
    const bool finished = introselect(b.begin(), b.begin() + nth, b.end(), comp, false);
    if (!finished) {                       /* what retain_best does at the depth limit: libstdc++ ran __heap_select there */
      deep++;
      return;                              /* the permutation of that case is checked through retain_best on the device tests */
    }
    tot++;
    if (a != b) bad++;
  };
  for (int t = 0; t < 20000; t++) {
    const int n = 4 + rng() % 300, ties = 1 + rng() % 40;
    std::vector<std::pair<float, int>> a(n);
    for (int i = 0; i < n; i++) a[i] = {(float)(rng() % ties), i};
    run(a, rng() % n);
  }
  for (int t = 0; t < 300; t++) {          /* level-sized inputs, integer scores with long runs of equal keys */
    const int n = 2000 + rng() % 6000, ties = 2 + rng() % 60;
    std::vector<std::pair<float, int>> a(n);
    for (int i = 0; i < n; i++) a[i] = {(float)(rng() % ties), i};
    run(a, std::min(n - 1, 100 + (int)(rng() % 1500)));
  }
  printf("mismatches %ld of %ld, depth-limit cases %ld\n", bad, tot, deep);
  return bad == 0 ? 0 : 1;
}
