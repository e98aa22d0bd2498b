/*
 * CPU oracle for the per-snapshot GCN message-passing path of spiro.dtdg.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under spiral_amd/ may import, link or call
 * this file; it is the checker used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * It restates, in plain C, what the reference computes on this path.  The
 * arithmetic itself lives in DGL 0.8.0.post1 (pinned at poetry.lock:206-207 of
 * the reference), which is not vendored in the reference tree; the functions
 * below follow DGL's published semantics at the reference's call sites:
 *
 *   orc_degree          DGLGraph.in_degrees()/out_degrees()
 *                       (spiro/dtdg/types.py:241; inside GraphConv.forward called
 *                       at .../implicitTimeEncoder/staticGraphEncoder.py:113,116)
 *   orc_coo_to_csr      dgl.graph((src,dst)) + lazy COOToCSR (types.py:191):
 *                       rows grouped by key, original edge order kept inside a
 *                       row (stable counting sort), eid = original position
 *   orc_norm            degs.float().clamp(min=1) ** -0.5  |  1/degs
 *   orc_aggregate       feat*norm ; update_all(copy_u|u_mul_e, sum) ; *norm ;
 *                       + bias ; activation  (GraphConv.forward, same call sites)
 *                       summed sequentially in CSR order per row in fp32, the
 *                       order DGL's CPU SpMMCsr uses; rows are OpenMP-parallel
 *                       like DGL's kernel.
 *
 * Pinning: the reference holds no numeric golden for the aggregation (its tests
 * assert counts and the in-degree feature sum on one 5-node fixture,
 * tests/dtdg/test_types_torch.py:36-56).  The degree/CSR functions are checked
 * against that fixture; the aggregation is "parity unpinned" by the reference
 * and is cross-checked against two independent restatements (numpy index-add
 * and scipy CSR) in tests/test_oracle.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off keeps every multiply and add separately rounded, as in the
 * reference's torch/DGL fp32 arithmetic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_ACT_NONE 0
#define ORC_ACT_RELU 1
#define ORC_ACT_LEAKY 2

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* deg[v] = number of occurrences of v in ids; returns -1 on an id outside [0,N) */
int orc_degree(const int64_t* ids, int64_t E, int64_t N, int64_t* deg) {
  memset(deg, 0, sizeof(int64_t) * (size_t)N);
  for (int64_t i = 0; i < E; ++i) {
    if (ids[i] < 0 || ids[i] >= N) return -1;
    deg[ids[i]] += 1;
  }
  return 0;
}

/* stable counting sort of the edge list by key */
int orc_coo_to_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N,
                   int64_t N_other, int32_t* indptr, int32_t* indices, int32_t* eid) {
  int64_t* cursor = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
  if (!cursor) return -2;
  for (int64_t i = 0; i < E; ++i) {
    if (key[i] < 0 || key[i] >= N || other[i] < 0 || other[i] >= N_other) {
      free(cursor);
      return -1;
    }
    cursor[key[i] + 1] += 1;
  }
  for (int64_t v = 0; v < N; ++v) cursor[v + 1] += cursor[v];
  for (int64_t v = 0; v <= N; ++v) indptr[v] = (int32_t)cursor[v];
  for (int64_t i = 0; i < E; ++i) {
    int64_t pos = cursor[key[i]]++;
    indices[pos] = (int32_t)other[i];
    eid[pos] = (int32_t)i;
  }
  free(cursor);
  return 0;
}

void orc_norm(const int32_t* indptr, int64_t N, int mode, float* norm) {
  for (int64_t v = 0; v < N; ++v) {
    float d = (float)(indptr[v + 1] - indptr[v]);
    if (d < 1.0f) d = 1.0f;
    norm[v] = (mode == 0) ? (1.0f / sqrtf(d)) : (1.0f / d);
  }
}

static inline float orc_act(float y, int act, float slope) {
  if (act == ORC_ACT_RELU) return y > 0.0f ? y : 0.0f;
  if (act == ORC_ACT_LEAKY) return y > 0.0f ? y : y * slope;
  return y;
}

/*
 * Y[r,:] = act( dst_scale[r] * sum_j (X[indices[j],:] * src_scale[indices[j]]) * edge_w[j] + bias )
 * with every product rounded to fp32 before the add (DGL scales the feature
 * matrix first, then sums messages).  threads <= 0 uses the OpenMP default.
 */
void orc_aggregate(const int32_t* indptr, const int32_t* indices, int64_t n_rows, const float* X,
                   int64_t ldx, int64_t F, const float* src_scale, const float* edge_w,
                   const float* dst_scale, const float* bias, int act, float slope, float* Y,
                   int64_t ldy, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  (void)threads;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t r = 0; r < n_rows; ++r) {
    float* y = Y + r * ldy;
    for (int64_t f = 0; f < F; ++f) y[f] = 0.0f;
    for (int32_t j = indptr[r]; j < indptr[r + 1]; ++j) {
      const float* x = X + (int64_t)indices[j] * ldx;
      const float ss = src_scale ? src_scale[indices[j]] : 1.0f;
      if (edge_w) {
        const float w = edge_w[j];
        for (int64_t f = 0; f < F; ++f) {
          float m = x[f] * ss;
          m = m * w;
          y[f] += m;
        }
      } else if (src_scale) {
        for (int64_t f = 0; f < F; ++f) {
          float m = x[f] * ss;
          y[f] += m;
        }
      } else {
        for (int64_t f = 0; f < F; ++f) y[f] += x[f];
      }
    }
    const float ds = dst_scale ? dst_scale[r] : 1.0f;
    for (int64_t f = 0; f < F; ++f) {
      float v = y[f];
      if (dst_scale) v = v * ds;
      if (bias) v = v + bias[f];
      y[f] = orc_act(v, act, slope);
    }
  }
}
