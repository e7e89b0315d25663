// Look-ahead word language model fusion kernels for gfx950 (HBM / latency bound, one workgroup per hypothesis).
//
// Reference: espresso/models/tensorized_lookahead_language_model.py:83-269 (Hori et al. 2018, "End-to-end speech
// recognition with word-based RNN language models", Eqn. 15, adapted to sentences that end with <space> <eos>) over the
// tensorized lexical prefix tree of espresso/tools/tensorized_prefix_tree.py:14-108
//   children[node][i]      node id of the i-th child (0 = "none": out of the tree), sorted by sub-word id
//   prev_subword_idx[node] sub-word on the edge into the node
//   word_idx[node]         word id if the node ends a word, else -1
//   word_set_idx[node]     (first-1, last) range of word ids sharing the prefix
// Per decoding step and hypothesis the reference issues ~40 small tensor ops (gathers, scatter, where, cumsum); here:
//   ea_softmax_cumsum      P(word | history) as an inclusive prefix sum over the (lexically sorted) word vocabulary
//   ea_lookahead_advance   automaton transition of every hypothesis on its last sub-word
//   ea_lookahead_logprobs  the sub-word log-probabilities of Eqn. 15 (all four cases) in one pass
#include "common.h"
#include "espresso_amd.h"

namespace {

// cumsum[n][v] = sum_{w <= v} softmax(logits[n])[w];  lp_tok[n] = log softmax(logits[n])[tok].  Rows with row_mask[n] == 0
// are left untouched.  One 256-thread workgroup per row; thread t owns the contiguous segment [t*seg, (t+1)*seg).
__global__ __launch_bounds__(256) void softmax_cumsum_kernel(const float* __restrict__ logits, long ld, const uint8_t* __restrict__ row_mask,
                                                             float* __restrict__ cumsum, float* __restrict__ lp_tok, int V, int tok) {
  __shared__ float sm[16];
  __shared__ float part[256];
  const int n = blockIdx.x;
  if (row_mask && !row_mask[n]) return;
  const float* z = logits + (long)n * ld;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, z[v]);
  mx = block_max(mx, sm);
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) s += expf(z[v] - mx);
  s = block_sum(s, sm);
  const float inv = 1.f / s;
  const int seg = (V + 255) / 256;
  const int v0 = threadIdx.x * seg, v1 = min(V, v0 + seg);
  float loc = 0.f;
  for (int v = v0; v < v1; ++v) loc += expf(z[v] - mx) * inv;
  part[threadIdx.x] = loc;
  __syncthreads();
  // exclusive scan of the 256 partial sums (Hillis-Steele in LDS)
  for (int off = 1; off < 256; off <<= 1) {
    const float add = threadIdx.x >= off ? part[threadIdx.x - off] : 0.f;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  float run = part[threadIdx.x] - loc;
  float* c = cumsum + (long)n * V;
  for (int v = v0; v < v1; ++v) {
    run += expf(z[v] - mx) * inv;
    c[v] = run;
  }
  if (threadIdx.x == 0 && lp_tok) lp_tok[n] = z[tok] - mx - logf(s);
}

__global__ __launch_bounds__(256) void lookahead_advance_kernel(int* __restrict__ nodes, const int* __restrict__ prev_tok,
                                                                const int* __restrict__ children, const int* __restrict__ prev_subword,
                                                                int N, int D, int space_idx, int root_id) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int tk = prev_tok[n];
  if (tk == space_idx) {  // inter-word transition: back to the root
    nodes[n] = root_id;
    return;
  }
  const int* ch = children + (long)nodes[n] * D;
  int nxt = 0;  // intra-word transition to the matching child, "none" (0) when the sub-word leaves the tree
  for (int i = 0; i < D; ++i) {
    const int c = ch[i];
    if (prev_subword[c] == tk) nxt += c;
  }
  nodes[n] = nxt;
}

struct LookaheadArgs {
  const int* nodes; const int* prev_tok; const float* cumsum; const float* lp_word_eos;
  const int* children; const int* prev_subword; const int* word_idx; const int* word_set;
  float* out;
  int N, Vw, Vs, D;
  float oov_penalty, zero;
  int open_vocab, word_unk, sub_space, sub_eos, sub_pad, none_id, root_id;
};

__global__ __launch_bounds__(64) void lookahead_logprobs_kernel(const LookaheadArgs a) {
  extern __shared__ float p[];  // [Vs]
  const int n = blockIdx.x;
  const int node = a.nodes[n];
  const int tk = a.prev_tok[n];
  const float* c = a.cumsum + (long)n * a.Vw;
  const bool space = tk == a.sub_space;
  const bool space_or_eos = space || tk == a.sub_eos;
  // cases 3 / 4 of Eqn. 15: OOV back-off mass everywhere, or probability 1 once the hypothesis has left the tree
  float base = a.zero;
  if (a.open_vocab) base = node == a.none_id ? 1.f : a.oov_penalty * (c[a.word_unk] - c[a.word_unk - 1]);
  for (int v = threadIdx.x; v < a.Vs; v += 64) {
    float x = base;
    if (a.open_vocab && node != a.none_id) {
      if (v == a.sub_space && space_or_eos) x = a.zero;
      if (v == a.sub_eos && !space) x = a.zero;
    }
    p[v] = x;
  }
  __syncthreads();
  float sum_probs = 1.f;
  if (node != a.none_id && node != a.root_id) sum_probs = c[a.word_set[2 * node + 1]] - c[a.word_set[2 * node]];
  // case 2: transitions to the children, P(words under the child) / P(words under this node)
  const int* ch = a.children + (long)node * a.D;
  for (int i = threadIdx.x; i < a.D; i += 64) {
    const int cn = ch[i];
    float val = (c[a.word_set[2 * cn + 1]] - c[a.word_set[2 * cn]]) / sum_probs;
    if (sum_probs < a.zero) val = a.zero;
    if (cn != a.none_id) p[a.prev_subword[cn]] = val;  // padded child slots all alias the pad column, cleared below
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    p[a.sub_pad] = a.zero;
    // case 1: the node ends a word -> <space> carries the word probability
    const int w = a.word_idx[node];
    if (w >= 0) p[a.sub_space] = sum_probs < a.zero ? a.zero : (c[w] - c[w - 1]) / sum_probs;
  }
  __syncthreads();
  float* o = a.out + (long)n * a.Vs;
  for (int v = threadIdx.x; v < a.Vs; v += 64) {
    float lp = logf(fmaxf(p[v], a.zero));
    if (v == a.sub_eos && space) lp = a.lp_word_eos[n];  // sentence end: the word LM's own <eos> probability
    o[v] = lp;
  }
}

}  // namespace

extern "C" int ea_softmax_cumsum(const float* logits, long ld, const uint8_t* row_mask, float* cumsum, float* lp_tok, int N, int V,
                                 int tok, hipStream_t stream) {
  if (N <= 0 || V <= 0) return 0;
  if (tok < 0 || tok >= V) return -2;
  hipLaunchKernelGGL(softmax_cumsum_kernel, dim3(N), dim3(256), 0, stream, logits, ld, row_mask, cumsum, lp_tok, V, tok);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_lookahead_advance(int* nodes, const int* prev_tok, const int* children, const int* prev_subword, int N, int D,
                                    int space_idx, int root_id, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(lookahead_advance_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, nodes, prev_tok, children, prev_subword, N, D,
                     space_idx, root_id);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_lookahead_logprobs(const int* nodes, const int* prev_tok, const float* cumsum, const float* lp_word_eos,
                                     const int* children, const int* prev_subword, const int* word_idx, const int* word_set, float* out,
                                     int N, int Vw, int Vs, int D, float oov_penalty, int open_vocab, int word_unk, int sub_space,
                                     int sub_eos, int sub_pad, hipStream_t stream) {
  if (N <= 0) return 0;
  if (word_unk < 1 || (size_t)Vs * sizeof(float) > 60 * 1024) return -2;
  LookaheadArgs a;
  a.nodes = nodes; a.prev_tok = prev_tok; a.cumsum = cumsum; a.lp_word_eos = lp_word_eos;
  a.children = children; a.prev_subword = prev_subword; a.word_idx = word_idx; a.word_set = word_set;
  a.out = out;
  a.N = N; a.Vw = Vw; a.Vs = Vs; a.D = D;
  a.oov_penalty = oov_penalty; a.zero = 1e-10f;
  a.open_vocab = open_vocab; a.word_unk = word_unk; a.sub_space = sub_space; a.sub_eos = sub_eos; a.sub_pad = sub_pad;
  a.none_id = 0; a.root_id = 1;
  hipLaunchKernelGGL(lookahead_logprobs_kernel, dim3(N), dim3(64), (size_t)Vs * sizeof(float), stream, a);
  return EA_CHECK_LAUNCH();
}
