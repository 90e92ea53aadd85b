// Throughput path: B independent dense LCQPs generated in HBM and solved by one launch of the homotopy kernel.
//   batch_synthetic [B=1024] [nV=256] [nC=512] [nComp=64]
#include <chrono>
#include <cstdio>
#include <cstdlib>

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
    return ok == B ? 0 : 1;
}
