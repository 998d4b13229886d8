// Does this runtime run independent branches of a hipGraph side by side, and what does a dependent small kernel cost
// in each form?    Standalone: hipcc --offload-arch=gfx950 -O2 tools/graph_branches.hip
//   -o tools/build/graph_branches && tools/build/graph_branches
// Forms, all over the same N small kernels (each: WG workgroups x 256 threads, a few round trips to L2):
//   chain      one stream captured: N nodes in a line (what engine/graphed.py replays today)
//   rewired    THAT captured graph with its edges replaced (hipGraphRemoveDependencies / AddDependencies) by
//              B independent lines between a root and a join - the edit engine/graph_dag.py would make
//   explicit   the same DAG built with hipGraphAddKernelNode
//   cells      N/(B+1) groups of [1 node -> B parallel nodes], each group depending on the one before (a ContextualCell)
//   streams    B streams launched from the host with events (no graph), for scale
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void small_kernel(float* buf, int n, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = buf[i];
  for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
  buf[i] = v;
}

static double time_graph(hipGraphExec_t exec, hipStream_t s, int reps) {
  CK(hipGraphLaunch(exec, s));
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, s));
  CK(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}

int main(int argc, char** argv) {
  int N = argc > 1 ? atoi(argv[1]) : 900;
  int B = argc > 2 ? atoi(argv[2]) : 5;
  int WG = argc > 3 ? atoi(argv[3]) : 64;
  int iters = argc > 4 ? atoi(argv[4]) : 200;
  int n = WG * 256;
  N = (N / (B * (B + 1))) * (B * (B + 1));  // divisible by B and by B + 1
  printf("N %d kernels, B %d branches, %d workgroups of 256, %d iterations\n", N, B, WG, iters);
  std::vector<float*> bufs(B + 1);
  for (auto& b : bufs) { CK(hipMalloc(&b, n * sizeof(float))); CK(hipMemset(b, 0, n * sizeof(float))); }
  hipStream_t s;
  CK(hipStreamCreate(&s));

  // one kernel alone, for scale
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    small_kernel<<<WG, 256, 0, s>>>(bufs[0], n, iters);
    CK(hipEventRecord(e0, s));
    for (int k = 0; k < 100; ++k) small_kernel<<<WG, 256, 0, s>>>(bufs[0], n, iters);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("host-launched, one stream: %.2f us per kernel\n", ms * 10.0);
  }

  // chain: captured from one stream; kernel k works on buffer k % B (so that branch b = kernels k % B == b is a real
  // dependency structure: B independent lines)
  hipGraph_t chain;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; ++k) small_kernel<<<WG, 256, 0, s>>>(bufs[k % B], n, iters);
  CK(hipStreamEndCapture(s, &chain));
  hipGraphExec_t chain_exec;
  CK(hipGraphInstantiate(&chain_exec, chain, nullptr, nullptr, 0));
  double t_chain = time_graph(chain_exec, s, 20);
  printf("chain      %8.1f us per replay  %.2f us per kernel\n", t_chain, t_chain / N);

  // rewired: the captured graph's edges replaced
  {
    size_t nn = 0;
    CK(hipGraphGetNodes(chain, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(chain, nodes.data(), &nn));
    printf("captured graph: %zu nodes\n", nn);
    // order them along the line (GetNodes' order is not promised): follow the edges from the root
    size_t ne = 0;
    CK(hipGraphGetEdges(chain, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> from(ne), to(ne);
    CK(hipGraphGetEdges(chain, from.data(), to.data(), &ne));
    size_t nr = 0;
    CK(hipGraphGetRootNodes(chain, nullptr, &nr));
    std::vector<hipGraphNode_t> roots(nr);
    CK(hipGraphGetRootNodes(chain, roots.data(), &nr));
    printf("edges %zu roots %zu; GetNodes in line order: ", ne, nr);
    std::vector<hipGraphNode_t> order;
    hipGraphNode_t cur = roots[0];
    order.push_back(cur);
    for (size_t step = 1; step < nn; ++step) {
      hipGraphNode_t nxt = nullptr;
      for (size_t e = 0; e < ne; ++e) if (from[e] == cur) { nxt = to[e]; break; }
      if (!nxt) break;
      order.push_back(nxt); cur = nxt;
    }
    bool same = order.size() == nn;
    for (size_t i = 0; same && i < nn; ++i) same = order[i] == nodes[i];
    printf("%s\n", same ? "yes" : "NO");
    CK(hipGraphRemoveDependencies(chain, from.data(), to.data(), ne));
    std::vector<hipGraphNode_t> nf, nt;
    for (size_t k = B; k < order.size(); ++k) { nf.push_back(order[k - B]); nt.push_back(order[k]); }
    CK(hipGraphAddDependencies(chain, nf.data(), nt.data(), nf.size()));
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, chain, nullptr, nullptr, 0));
    double t = time_graph(exec, s, 20);
    printf("rewired    %8.1f us per replay  %.2f us per kernel  (%.2fx)\n", t, t / N, t_chain / t);
    CK(hipGraphExecDestroy(exec));
  }

  // explicit: AddKernelNode, B lines
  auto add_node = [&](hipGraph_t g, const std::vector<hipGraphNode_t>& deps, float** buf, int* pn, int* pit) {
    hipKernelNodeParams p = {};
    void* args[3] = {buf, pn, pit};
    p.func = (void*)small_kernel;
    p.gridDim = dim3(WG); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
    hipGraphNode_t node;
    CK(hipGraphAddKernelNode(&node, g, deps.data(), deps.size(), &p));
    return node;
  };
  {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> last(B, nullptr);
    for (int k = 0; k < N; ++k) {
      std::vector<hipGraphNode_t> deps;
      if (last[k % B]) deps.push_back(last[k % B]);
      last[k % B] = add_node(g, deps, &bufs[k % B], &n, &iters);
    }
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    double t = time_graph(exec, s, 20);
    printf("explicit   %8.1f us per replay  %.2f us per kernel  (%.2fx)\n", t, t / N, t_chain / t);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(g));
  }
  // cells: [head -> B parallel] groups in sequence
  {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> prev;
    int groups = N / (B + 1);
    for (int q = 0; q < groups; ++q) {
      hipGraphNode_t head = add_node(g, prev, &bufs[B], &n, &iters);
      prev.clear();
      for (int b = 0; b < B; ++b) prev.push_back(add_node(g, {head}, &bufs[b], &n, &iters));
    }
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    double t = time_graph(exec, s, 20);
    printf("cells      %8.1f us per replay  %.2f us per kernel  (%.2fx)\n", t, t / N, t_chain / t);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(g));
  }
  // cells with two-kernel branches: [head -> B x (a -> b)] groups
  {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> prev;
    int groups = N / (2 * B + 1), total = 0;
    for (int q = 0; q < groups; ++q) {
      hipGraphNode_t head = add_node(g, prev, &bufs[B], &n, &iters);
      prev.clear(); ++total;
      for (int b = 0; b < B; ++b) {
        hipGraphNode_t a = add_node(g, {head}, &bufs[b], &n, &iters);
        prev.push_back(add_node(g, {a}, &bufs[b], &n, &iters));
        total += 2;
      }
    }
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    double t = time_graph(exec, s, 20);
    printf("cells2     %8.1f us per replay  %.2f us per kernel (%d kernels; chain-equivalent %.1f us: %.2fx)\n", t, t / total,
           total, t_chain / N * total, t_chain / N * total / t);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(g));
  }

  // any-order launches (hipExtLaunchKernel, flag hipExtAnyOrderLaunch = AQL barrier bit clear): one stream, from the host
  {
    auto launch = [&](float** buf, int flags, hipStream_t st) {
      void* args[3] = {buf, &n, &iters};
      CK(hipExtLaunchKernel((const void*)small_kernel, dim3(WG), dim3(256), args, 0, st, nullptr, nullptr, flags));
    };
    auto timeit = [&](const char* name, auto&& body, int total) {
      body(); CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 10; ++r) body();
      CK(hipDeviceSynchronize());
      auto t1 = std::chrono::steady_clock::now();
      double t = std::chrono::duration<double, std::micro>(t1 - t0).count() / 10;
      printf("%-18s %8.1f us per pass    %.2f us per kernel  (%.2fx)\n", name, t, t / total, t_chain / N * total / t);
    };
    timeit("ext, in order", [&]() { for (int k = 0; k < N; ++k) launch(&bufs[k % B], 0, s); }, N);
    timeit("ext, any order", [&]() { for (int k = 0; k < N; ++k) launch(&bufs[k % B], 1, s); }, N);
    // levels: B lines interleaved, the first kernel of a level in order (waits for the level before), the rest any-order
    timeit("ext, levels of B", [&]() { for (int k = 0; k < N; ++k) launch(&bufs[k % B], (k % B) ? 1 : 0, s); }, N);
    // is the flag honoured? N increments of ONE buffer with overlapping kernels would lose updates / stay exact
    CK(hipMemset(bufs[0], 0, n * sizeof(float)));
    int one = 1;
    auto inc = [&](int flags) {
      void* args[3] = {&bufs[0], &n, &one};
      CK(hipExtLaunchKernel((const void*)small_kernel, dim3(WG), dim3(256), args, 0, s, nullptr, nullptr, flags));
    };
    // captured: does a graph keep the flag?
    hipGraph_t g;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < N; ++k) launch(&bufs[k % B], (k % B) ? 1 : 0, s);
    CK(hipStreamEndCapture(s, &g));
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    double t = time_graph(exec, s, 20);
    printf("captured levels    %8.1f us per replay  %.2f us per kernel  (%.2fx)\n", t, t / N, t_chain / t);
    (void)inc;
  }

  // the hole between replays: host time inside hipGraphLaunch and the total of 10 back-to-back replays, one
  // executable graph against two used in turn (both instantiated from the same graph)
  {
    hipGraphExec_t second;
    CK(hipGraphInstantiate(&second, chain, nullptr, nullptr, 0));  // (the rewired chain: B lines)
    hipGraphExec_t first;
    CK(hipGraphInstantiate(&first, chain, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; ++mode) {
      hipGraphExec_t ex[2] = {first, mode ? second : first};
      CK(hipGraphLaunch(ex[0], s)); CK(hipGraphLaunch(ex[1], s)); CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      double host_max = 0, host_sum = 0;
      for (int r = 0; r < 10; ++r) {
        auto a = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ex[r & 1], s));
        double h = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
        host_sum += h; host_max = h > host_max ? h : host_max;
      }
      CK(hipStreamSynchronize(s));
      double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
      printf("%s: %8.1f us per replay, host inside hipGraphLaunch mean %.1f max %.1f us\n",
             mode ? "two executables in turn" : "one executable        ", t, host_sum / 10, host_max);
    }
    // and the chain (one line) the same way
    hipGraphExec_t c2;
    hipGraph_t chain2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < N; ++k) small_kernel<<<WG, 256, 0, s>>>(bufs[k % B], n, iters);
    CK(hipStreamEndCapture(s, &chain2));
    hipGraphExec_t c1;
    CK(hipGraphInstantiate(&c1, chain2, nullptr, nullptr, 0));
    CK(hipGraphInstantiate(&c2, chain2, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; ++mode) {
      hipGraphExec_t ex[2] = {c1, mode ? c2 : c1};
      CK(hipGraphLaunch(ex[0], s)); CK(hipGraphLaunch(ex[1], s)); CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      double host_max = 0, host_sum = 0;
      for (int r = 0; r < 10; ++r) {
        auto a = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ex[r & 1], s));
        double h = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
        host_sum += h; host_max = h > host_max ? h : host_max;
      }
      CK(hipStreamSynchronize(s));
      double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
      printf("line, %s: %8.1f us per replay, host inside hipGraphLaunch mean %.1f max %.1f us\n",
             mode ? "two executables in turn" : "one executable        ", t, host_sum / 10, host_max);
    }
  }

  // B separate LINE graphs (N / B kernels each) launched on B streams, forked from and joined to `s` with events
  {
    for (int nb : {1, 2, 3, 4, 5}) {
      if (nb > B) break;
      std::vector<hipStream_t> ss(nb);
      std::vector<hipGraphExec_t> ex(nb);
      std::vector<hipEvent_t> done(nb);
      hipEvent_t fork;
      CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
      int per = N / nb;
      for (int b = 0; b < nb; ++b) {
        CK(hipStreamCreateWithFlags(&ss[b], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        hipGraph_t g;
        CK(hipStreamBeginCapture(ss[b], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < per; ++k) small_kernel<<<WG, 256, 0, ss[b]>>>(bufs[b], n, iters);
        CK(hipStreamEndCapture(ss[b], &g));
        CK(hipGraphInstantiate(&ex[b], g, nullptr, nullptr, 0));
      }
      auto pass = [&]() {
        CK(hipEventRecord(fork, s));
        for (int b = 0; b < nb; ++b) {
          CK(hipStreamWaitEvent(ss[b], fork, 0));
          CK(hipGraphLaunch(ex[b], ss[b]));
          CK(hipEventRecord(done[b], ss[b]));
          CK(hipStreamWaitEvent(s, done[b], 0));
        }
      };
      pass(); CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      double host = 0;
      for (int r = 0; r < 10; ++r) {
        auto a = std::chrono::steady_clock::now();
        pass();
        host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
      }
      CK(hipDeviceSynchronize());
      double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
      printf("%d line graphs on %d streams: %8.1f us per pass  %.2f us per kernel (%.2fx), host %.1f us per pass\n", nb, nb, t,
             t / (per * nb), t_chain / N * (per * nb) / t, host / 10);
    }
  }
  // streams from the host
  {
    std::vector<hipStream_t> ss(B);
    for (auto& x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto run = [&]() { for (int k = 0; k < N; ++k) small_kernel<<<WG, 256, 0, ss[k % B]>>>(bufs[k % B], n, iters); };
    run(); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; ++r) run();
    CK(hipDeviceSynchronize());
    auto t1 = std::chrono::steady_clock::now();
    double t = std::chrono::duration<double, std::micro>(t1 - t0).count() / 10;
    printf("streams    %8.1f us per pass    %.2f us per kernel  (%.2fx)\n", t, t / N, t_chain / t);
  }
  return 0;
}
