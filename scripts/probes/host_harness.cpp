#include "lfr.h"
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
    for (int rep = 0; rep < 3; ++rep) {
        lfr_graph *g = nullptr; lfr_problem *p = nullptr;
        if (lfr_graph_from_matches_file(argv[1], nullptr, 0, &g) != 0) { fprintf(stderr, "parse: %s\n", lfr_last_error()); return 1; }
        if (lfr_problem_build(g, 0, nullptr, &p) != 0) { fprintf(stderr, "build: %s\n", lfr_last_error()); return 1; }
        lfr_problem_stats st; lfr_problem_get_stats(p, &st);
        printf("nodes %lld tracks %lld comps %lld cut %lld\n", (long long)lfr_graph_num_nodes(g), (long long)st.n_tracks, (long long)st.n_components, (long long)st.n_cut_components);
        lfr_problem_free(p); lfr_graph_free(g);
    }
    {   // the persistent host workers: one caller after the other, then four at once (three of them fall back to threads of their own)
        long long total = lfr_debug_pool_selftest(32, 20000, 10);
        std::vector<std::thread> th;
        long long part[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) th.emplace_back([&part, i] { part[i] = lfr_debug_pool_selftest(8 + 4 * i, 5000, 20); });
        for (auto &t : th) t.join();
        printf("pool: %lld + %lld %lld %lld %lld\n", total, part[0], part[1], part[2], part[3]);
    }
    return 0;
}
