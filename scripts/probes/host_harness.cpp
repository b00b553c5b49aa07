#include "lfr.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) {
    for (int rep = 0; rep < 3; ++rep) {
        lfr_graph *g = nullptr; lfr_problem *p = nullptr;
        if (lfr_graph_from_matches_file(argv[1], nullptr, 0, &g) != 0) { fprintf(stderr, "parse: %s\n", lfr_last_error()); return 1; }
        if (lfr_problem_build(g, 0, nullptr, &p) != 0) { fprintf(stderr, "build: %s\n", lfr_last_error()); return 1; }
        lfr_problem_stats st; lfr_problem_get_stats(p, &st);
        printf("nodes %lld tracks %lld comps %lld cut %lld\n", (long long)lfr_graph_num_nodes(g), (long long)st.n_tracks, (long long)st.n_components, (long long)st.n_cut_components);
        lfr_problem_free(p); lfr_graph_free(g);
    }
    return 0;
}
