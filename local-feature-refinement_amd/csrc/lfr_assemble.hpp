// Device-side batch assembly (lfr_assemble.hip): builds the HBM batch layout of a whole problem on
// the GPU from the match graph + the host graph stage's labels.
#pragma once
#include <hip/hip_runtime.h>

#include "lfr_internal.hpp"

namespace lfr {

struct DeviceAssembly {
    // device arrays (handed over to the batch)
    CompDesc *d_descs = nullptr;
    EdgeRec *d_edges = nullptr;
    uint32_t *d_node_ids = nullptr;
    NodeInc *d_node_inc = nullptr;
    uint32_t *d_in_idx = nullptr;
    // host mirrors
    std::vector<CompDesc> descs;
    std::vector<int64_t> desc_component;
    std::vector<int32_t> desc_class, desc_tracks;
    std::vector<uint32_t> node_ids;
    int64_t n_edges = 0, n_nodes = 0;
    void release();
};

// dev_disp1/dev_disp2: optional device pointers to the flows in match order (n_matches x 18 floats
// each, disp1 = flow 2->1, disp2 = flow 1->2); when null the graph's host arrays are uploaded.
// One slab per pipeline stage: hipFree costs ~0.2 ms (it synchronises the device) and each stage uses ~40
// temporaries.  A buffer that does not fit falls back to its own hipMalloc/hipFree.
struct DevArena {
    char *base = nullptr;
    size_t cap = 0, top = 0;
    ~DevArena() { if (base) (void)hipFree(base); }
    hipError_t init(size_t bytes) { cap = bytes; return hipMalloc((void **)&base, bytes); }
    void *take(size_t bytes) {
        const size_t b = (bytes + 255) & ~(size_t)255;
        if (!base || top + b > cap) return nullptr;
        void *p = base + top;
        top += b;
        return p;
    }
};
struct DevBuf {          // tiny RAII for the many temporaries
    void *p = nullptr;
    DevArena *arena = nullptr;
    size_t mark = 0;
    bool temp = false;       // released at scope end (stack discipline: nothing is taken while it lives)
    ~DevBuf() {
        if (!p) return;
        if (!arena) (void)hipFree(p);
        else if (temp) arena->top = mark;
    }
    template <class T> T *as() { return (T *)p; }
};
inline hipError_t dev_alloc(DevArena *ar, DevBuf &b, size_t bytes, bool temp) {
    bytes = bytes < 16 ? 16 : bytes;
    if (ar) {
        const size_t m = ar->top;
        if (void *q = ar->take(bytes)) { b.p = q; b.arena = ar; b.mark = m; b.temp = temp; return hipSuccess; }
    }
    return hipMalloc(&b.p, bytes);
}

int assemble_on_device(const Graph &g, const Problem &labels, hipStream_t stream, const float *dev_disp1,
                       const float *dev_disp2, DeviceAssembly &out);

// Tracks, roots and components on the GPU (lfr_graphstage.hip): fills p.track / p.comp / p.is_root and
// the stage statistics exactly as the host stage does.  Returns LFR_GRAPHSTAGE_USE_HOST when the
// input needs something only the host stage has (graph cut above the size cap, a huge connected
// component): the caller then runs build_problem().
constexpr int LFR_GRAPHSTAGE_USE_HOST = 1;
int graph_stage_on_device(const Graph &g, int64_t max_nodes, int device, Problem &p);

}  // namespace lfr
