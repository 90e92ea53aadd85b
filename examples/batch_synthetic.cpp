// Throughput path: B independent dense LCQPs generated in HBM and solved by one launch of the homotopy kernel.
//   batch_synthetic [B=1024] [nV=256] [nC=512] [nComp=64] [pipelined steps=8]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "BatchLCQProblem.hpp"

using namespace LCQPow;

int main(int argc, char** argv)
{
    const int B = argc > 1 ? std::atoi(argv[1]) : 1024, nV = argc > 2 ? std::atoi(argv[2]) : 256;
    const int nC = argc > 3 ? std::atoi(argv[3]) : 512, nComp = argc > 4 ? std::atoi(argv[4]) : 64;
    BatchLCQProblem batch(B, nV, nC, nComp);
    if (!batch.ok()) { std::printf("could not create the batch: %s\n", lcqp_hip_last_error()); return 1; }
    Options options;
    options.setPrintLevel(NONE);
    options.setPerturbStep(false);
    batch.setOptions(options);
    batch.generateSynthetic(0x4C43515000000001ULL, 0);
    batch.runSolver();   // warm-up (first launch loads the code object)
    const auto t0 = std::chrono::steady_clock::now();
    if (batch.runSolver() != SUCCESSFUL_RETURN) { std::printf("runSolver failed: %s\n", lcqp_hip_last_error()); return 1; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int ok = 0;
    long iters = 0;
    for (int i = 0; i < B; i++) { ok += batch.getReturnValue(i) == SUCCESSFUL_RETURN; iters += batch.getStats(i).iterTotal; }
    std::printf("%d/%d LCQPs solved in %.1f ms (%.0f LCQPs/s), %.1f iterates per LCQP\n", ok, B, dt * 1e3, B / dt, (double)iters / B);
    if (ok != B) return 1;
    // the same work as a stream of batches over two batch objects (BatchPipeline): batch k+1 is generated and launched while batch k runs
    const int steps = argc > 5 ? std::atoi(argv[5]) : 8;
    BatchPipeline pipe(2, B, nV, nC, nComp);
    if (!pipe.ok()) { std::printf("could not create the pipeline: %s\n", lcqp_hip_last_error()); return 1; }
    for (int k = 0; k < pipe.depth(); k++) { pipe.slot(k).setOptions(options); pipe.slot(k).generateSynthetic(0x4C43515000000001ULL, 0); pipe.slot(k).runSolver(); }
    long solved = 0;
    double check = 0.0;
    std::vector<double> xv(nV);
    auto consume = [&](BatchLCQProblem& b) {
        for (int i = 0; i < B; i++) { solved += b.getReturnValue(i) == SUCCESSFUL_RETURN; }
        b.getPrimalSolution(0, xv.data()); check += xv[0];
    };
    const auto t1 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; k++) {
        BatchLCQProblem& b = pipe.acquire();
        if (pipe.hasResults()) consume(b);
        // (a real caller loads the next batch of problems here: loadLCQP / generateSynthetic)
        if (pipe.launch(b) != SUCCESSFUL_RETURN) { std::printf("launch failed: %s\n", lcqp_hip_last_error()); return 1; }
    }
    while (BatchLCQProblem* b = pipe.drain()) consume(*b);
    const double dp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    std::printf("pipelined, depth 2: %ld/%ld LCQPs solved in %.1f ms (%.0f LCQPs/s)\n", solved, (long)steps * B, dp * 1e3, steps * (double)B / dp);
    return solved == (long)steps * B ? 0 : 1;
}
