// Batch sharding from C++: contiguous slices of instance ids, one host thread and one BatchLCQProblem (its own HIP
// stream) per shard, shard s on device s % deviceCount, no exchange between shards; results are joined on the host.
// With fewer devices than shards several shards share a device -- which is also how the thread safety of the library is tested
// on a one-GPU box (tests/test_host_cpp.py).
//   multi_gpu_batch [total=2048] [shards=#devices] [nV=256] [nC=512] [nComp=64]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "BatchLCQProblem.hpp"

using namespace LCQPow;

struct ShardResult { int solved = 0; long iterates = 0; double checksum = 0.0; bool ok = false; };

int main(int argc, char** argv)
{
    lcqp_hip_request_hw_queues(8);      // before the first HIP call: one queue per shard stream (the library never sets it by itself)
    const int ndev = lcqp_hip_device_count();
    if (ndev < 1) { std::printf("no GPU visible\n"); return 1; }
    const int total = argc > 1 ? std::atoi(argv[1]) : 2048, shards = argc > 2 ? std::atoi(argv[2]) : ndev;
    const int nV = argc > 3 ? std::atoi(argv[3]) : 256, nC = argc > 4 ? std::atoi(argv[4]) : 512, nComp = argc > 5 ? std::atoi(argv[5]) : 64;
    if (shards < 1 || total % shards) { std::printf("total must be a multiple of shards\n"); return 1; }
    const int per = total / shards;
    std::vector<ShardResult> res(shards);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int s = 0; s < shards; s++)
        th.emplace_back([&, s]() {
            BatchLCQProblem batch(per, nV, nC, nComp, false, s % ndev);
            if (!batch.ok()) return;
            Options options;
            options.setPrintLevel(NONE);
            options.setPerturbStep(false);
            batch.setOptions(options);
            batch.generateSynthetic(0x4C43515000000001ULL, (unsigned long long)s * per);     // instance ids [s*per, (s+1)*per)
            if (batch.runSolver() != SUCCESSFUL_RETURN) return;
            std::vector<double> x(nV);
            for (int i = 0; i < per; i++) {
                res[s].solved += batch.getReturnValue(i) == SUCCESSFUL_RETURN;
                res[s].iterates += batch.getStats(i).iterTotal;
                batch.getPrimalSolution(i, x.data());
                for (int k = 0; k < nV; k++) res[s].checksum += x[k];
            }
            res[s].ok = true;
        });
    for (auto& t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int solved = 0; long iterates = 0; double checksum = 0.0; bool ok = true;
    for (const ShardResult& r : res) { solved += r.solved; iterates += r.iterates; checksum += r.checksum; ok = ok && r.ok; }
    std::printf("%d shards on %d device(s): %d/%d LCQPs solved, %.1f iterates per LCQP, checksum %.12e, %.1f ms incl. generation and setup\n",
                shards, ndev, solved, total, (double)iterates / total, checksum, dt * 1e3);
    return ok && solved == total ? 0 : 1;
}
