// BatchLCQProblem: B independent dense LCQPs of one shape solved entirely on one MI355X (one persistent
// workgroup per instance).  An extension behind the reference surface -- the reference has no batched
// API -- whose per-instance semantics are those of LCQProblem::loadLCQP / runSolver.
#ifndef LCQPOW_AMD_BATCHLCQPROBLEM_HPP
#define LCQPOW_AMD_BATCHLCQPROBLEM_HPP

#include <vector>

#include "Options.hpp"
#include "OutputStatistics.hpp"

namespace LCQPow {

class BatchLCQProblem {
  public:
    BatchLCQProblem(int batch, int nV, int nC, int nComp, bool withBoxBounds = false, int device = 0)
        : B(batch), nV_(nV), nC_(nC), nComp_(nComp), h(lcqp_hip_batch_create(batch, nV, nC, nComp, withBoxBounds ? 1 : 0, device)) {}
    ~BatchLCQProblem() { if (h) lcqp_hip_batch_destroy(h); }
    BatchLCQProblem(const BatchLCQProblem&) = delete;
    BatchLCQProblem& operator=(const BatchLCQProblem&) = delete;

    bool ok() const { return h != nullptr; }
    ReturnValue setOptions(const Options& o) { return (ReturnValue)lcqp_hip_batch_set_options(h, &o.getHIPOptions()); }
    // the setup of this object runs beside another object's homotopy kernel (BatchPipeline sets it): see lcqp_hip_batch_set_overlapped
    ReturnValue setOverlapped(bool overlapped) { return (ReturnValue)lcqp_hip_batch_set_overlapped(h, overlapped ? 1 : 0); }
    // instance-by-instance load with the argument list of LCQProblem::loadLCQP
    ReturnValue loadLCQP(int instance, const double* Q, const double* g, const double* L, const double* R,
                         const double* lbL = 0, const double* ubL = 0, const double* lbR = 0, const double* ubR = 0,
                         const double* A = 0, const double* lbA = 0, const double* ubA = 0, const double* lb = 0,
                         const double* ub = 0, const double* x0 = 0, const double* y0 = 0)
    {
        return (ReturnValue)lcqp_hip_batch_load(h, instance, 1, Q, g, L, R, lbL, ubL, lbR, ubR, A, lbA, ubA, lb, ub, x0, y0);
    }
    ReturnValue generateSynthetic(unsigned long long seed0, unsigned long long firstInstance)
    {
        return (ReturnValue)lcqp_hip_batch_generate_synthetic(h, seed0, firstInstance);
    }
    // runSolver for every instance; per-instance return values are in getReturnValue(i)
    ReturnValue runSolver()
    {
        const ReturnValue rc = runSolverAsync();
        return rc != SUCCESSFUL_RETURN ? rc : collect();
    }
    // the two halves of runSolver: the launches on this batch's own HIP stream (returns at once), and the wait + read-back.  Between the two
    // the host is free -- e.g. to load and launch another batch object (BatchPipeline below).
    ReturnValue runSolverAsync() { return (ReturnValue)lcqp_hip_batch_run(h); }
    ReturnValue collect()
    {
        x.assign((size_t)B * nV_, 0.0);
        y.assign((size_t)B * (nV_ + nC_ + 2 * nComp_), 0.0);
        st.assign(B, lcqp_stats_t());
        return (ReturnValue)lcqp_hip_batch_get_solution(h, x.data(), y.data(), st.data());
    }
    ReturnValue getReturnValue(int i) const { return (ReturnValue)st[i].returnValue; }
    AlgorithmStatus getPrimalSolution(int i, double* xOpt) const
    {
        for (int k = 0; k < nV_; ++k) xOpt[k] = x[(size_t)i * nV_ + k];
        return (AlgorithmStatus)st[i].status;
    }
    AlgorithmStatus getDualSolution(int i, double* yOpt) const
    {
        const int nd = nV_ + nC_ + 2 * nComp_;
        for (int k = 0; k < nd; ++k) yOpt[k] = y[(size_t)i * nd + k];
        return (AlgorithmStatus)st[i].status;
    }
    const lcqp_stats_t& getStats(int i) const { return st[i]; }
    int getNumberOfPrimals() const { return nV_; }
    int getNumberOfDuals() const { return nV_ + nC_ + 2 * nComp_; }
    lcqp_hip_batch_t* handle() { return h; }

  private:
    int B, nV_, nC_, nComp_;
    lcqp_hip_batch_t* h;
    std::vector<double> x, y;
    std::vector<lcqp_stats_t> st;
};

// A stream of batches over `depth` batch objects.  A launch whose batch fills the GPU exactly once ends with its
// slowest instances while most workgroup slots are already idle (B = 1024: an average slot is busy 80 % of the launch); with a second batch
// object in flight -- its own buffers, its own HIP stream -- the setup kernels and the first instances of batch k+1 run in that tail.  Every
// batch still does all of runSolver's work; results are bit-identical to the sequential use.  Usage:
//     BatchPipeline pipe(2, 1024, nV, nC, nComp);
//     for (each batch of problems) { BatchLCQProblem& b = pipe.acquire();   // waits for (and hands back) the oldest batch when all are in flight
//                                    if (b.hasResults()) consume(b);  load(b);  pipe.launch(b); }
//     while (BatchLCQProblem* b = pipe.drain()) consume(*b);
// (HIP maps the streams of a process onto a few hardware queues -- 4 by default, GPU_MAX_HW_QUEUES -- and every batch object has two streams:
//  a process that keeps more batch objects alive than the pipeline's can find both slots on one queue, and they then run one after the
//  other; tools/micro/pipeline_check.py measures it.  Create the pipeline first, or raise GPU_MAX_HW_QUEUES.)
class BatchPipeline {
  public:
    BatchPipeline(int depth, int batch, int nV, int nC, int nComp, bool withBoxBounds = false, int device = 0)
    {
        for (int k = 0; k < depth; ++k) {
            slots.push_back(new BatchLCQProblem(batch, nV, nC, nComp, withBoxBounds, device)); state.push_back(0);
            if (depth > 1 && slots.back()->ok()) slots.back()->setOverlapped(true);
        }
    }
    ~BatchPipeline() { for (size_t k = 0; k < slots.size(); ++k) delete slots[k]; }
    BatchPipeline(const BatchPipeline&) = delete;
    BatchPipeline& operator=(const BatchPipeline&) = delete;
    bool ok() const { for (size_t k = 0; k < slots.size(); ++k) if (!slots[k]->ok()) return false; return !slots.empty(); }
    int depth() const { return (int)slots.size(); }
    BatchLCQProblem& slot(int k) { return *slots[k]; }
    // the batch object to fill next: a free one, else the oldest one in flight (waited for; its results are then in the object: resultsOf())
    // (a slot that was handed out with results and not launched again -- a failed launch, or acquire() twice in a row -- is free again)
    BatchLCQProblem& acquire()
    {
        if (cur >= 0 && state[cur] == 2) state[cur] = 0;
        for (size_t k = 0; k < slots.size(); ++k) if (state[k] == 0) { cur = (int)k; return *slots[k]; }
        if (order.empty()) { for (size_t k = 0; k < slots.size(); ++k) state[k] = 0; cur = 0; return *slots[0]; }      // nothing in flight: every slot is free
        const int k = order.front(); order.erase(order.begin());
        lastRc = slots[k]->collect(); state[k] = 2; cur = k;
        return *slots[k];
    }
    bool hasResults() const { return cur >= 0 && state[cur] == 2; }     // the object acquire() returned carries a finished run
    ReturnValue lastCollectStatus() const { return lastRc; }
    ReturnValue launch(BatchLCQProblem& b)
    {
        for (size_t k = 0; k < slots.size(); ++k)
            if (slots[k] == &b) { const ReturnValue rc = b.runSolverAsync(); if (rc == SUCCESSFUL_RETURN) { state[k] = 1; order.push_back((int)k); } return rc; }
        return INVALID_ARGUMENT;
    }
    // after the last launch: the batches still in flight, oldest first (NULL when none is left)
    BatchLCQProblem* drain()
    {
        if (order.empty()) return 0;
        const int k = order.front(); order.erase(order.begin());
        lastRc = slots[k]->collect(); state[k] = 0;
        if (order.empty()) for (size_t q = 0; q < slots.size(); ++q) state[q] = 0;      // drained: a pipeline that is used again starts with every slot free
        return slots[k];
    }

  private:
    std::vector<BatchLCQProblem*> slots;
    std::vector<int> state;      // 0 free, 1 in flight, 2 finished (results in the object)
    std::vector<int> order;      // launch order of the batches in flight
    int cur = -1;
    ReturnValue lastRc = SUCCESSFUL_RETURN;
};

// A list of LCQPs of DIFFERENT shapes solved in one call (round 6).  The reference is one object per problem, any mix of sizes
// (include/LCQProblem.hpp:56-60, 87-144); the device kernels take one shape per batch object, so the problems are sorted into buckets of equal
// (nV, nC, nComp, box bounds or not) -- one BatchLCQProblem per bucket, created for exactly that shape, so that every instance gets the bits of
// its solo run -- and the buckets go through the device two at a time: bucket k + 1 is created, loaded and launched while bucket k runs
// (each batch object has its own HIP stream).  Within a bucket instances may differ in everything the reference lets differ per problem:
// lbL / lbR / ubL / ubR, lbA / ubA, lb / ub, x0, y0 given or not.
// Usage:   MixedBatchLCQProblem mixed;  mixed.setOptions(opt);
//          int i = mixed.addProblem(nV, nC, nComp, Q, g, L, R, ...);      // argument list of LCQProblem::loadLCQP; pointers are borrowed until runSolver returns
//          mixed.runSolver();  mixed.getPrimalSolution(i, x);  mixed.getReturnValue(i);
class MixedBatchLCQProblem {
  public:
    explicit MixedBatchLCQProblem(int device = 0) : device_(device), haveOptions(false) {}
    ReturnValue setOptions(const Options& o) { options = o; haveOptions = true; return SUCCESSFUL_RETURN; }
    int addProblem(int nV, int nC, int nComp, const double* Q, const double* g, const double* L, const double* R,
                   const double* lbL = 0, const double* ubL = 0, const double* lbR = 0, const double* ubR = 0,
                   const double* A = 0, const double* lbA = 0, const double* ubA = 0, const double* lb = 0,
                   const double* ub = 0, const double* x0 = 0, const double* y0 = 0)
    {
        Problem p = {nV, nC, nComp, Q, g, L, R, lbL, ubL, lbR, ubR, A, lbA, ubA, lb, ub, x0, y0, -1, -1};
        problems.push_back(p);
        return (int)problems.size() - 1;
    }
    int size() const { return (int)problems.size(); }
    int numberOfBuckets() const { return (int)buckets.size(); }
    // runSolver for every problem; the return value is that of the first call that failed to LOAD or LAUNCH (per-problem results: getReturnValue)
    ReturnValue runSolver()
    {
        for (size_t k = 0; k < buckets.size(); ++k) delete buckets[k].batch;
        buckets.clear();
        for (size_t i = 0; i < problems.size(); ++i) {
            Problem& p = problems[i];
            const bool box = p.lb || p.ub;
            size_t k = 0;
            for (; k < buckets.size(); ++k) if (buckets[k].nV == p.nV && buckets[k].nC == p.nC && buckets[k].nComp == p.nComp && buckets[k].box == box) break;
            if (k == buckets.size()) { Bucket b = {p.nV, p.nC, p.nComp, box, std::vector<int>(), 0}; buckets.push_back(b); }
            p.bucket = (int)k; p.slot = (int)buckets[k].members.size();
            buckets[k].members.push_back((int)i);
        }
        ReturnValue first = SUCCESSFUL_RETURN;
        int inFlight = -1;
        for (size_t k = 0; k < buckets.size(); ++k) {
            Bucket& b = buckets[k];
            b.batch = new BatchLCQProblem((int)b.members.size(), b.nV, b.nC, b.nComp, b.box, device_);
            ReturnValue rc = b.batch->ok() ? SUCCESSFUL_RETURN : LCQPOBJECT_NOT_SETUP;
            if (rc == SUCCESSFUL_RETURN && haveOptions) rc = b.batch->setOptions(options);
            if (rc == SUCCESSFUL_RETURN && buckets.size() > 1) rc = b.batch->setOverlapped(true);      // beside the bucket launched before it
            for (size_t s = 0; s < b.members.size() && rc == SUCCESSFUL_RETURN; ++s) {
                const Problem& p = problems[b.members[s]];
                rc = b.batch->loadLCQP((int)s, p.Q, p.g, p.L, p.R, p.lbL, p.ubL, p.lbR, p.ubR, p.A, p.lbA, p.ubA, p.lb, p.ub, p.x0, p.y0);
            }
            if (rc == SUCCESSFUL_RETURN) rc = b.batch->runSolverAsync();
            if (rc != SUCCESSFUL_RETURN) { if (first == SUCCESSFUL_RETURN) first = rc; delete b.batch; b.batch = 0; }
            // the bucket launched before this one is collected now: its kernels ran while this one was created and loaded
            if (inFlight >= 0) { const ReturnValue rcC = buckets[inFlight].batch->collect(); if (rcC != SUCCESSFUL_RETURN && first == SUCCESSFUL_RETURN) first = rcC; }
            inFlight = b.batch ? (int)k : -1;
        }
        if (inFlight >= 0) { const ReturnValue rcC = buckets[inFlight].batch->collect(); if (rcC != SUCCESSFUL_RETURN && first == SUCCESSFUL_RETURN) first = rcC; }
        return first;
    }
    ~MixedBatchLCQProblem() { for (size_t k = 0; k < buckets.size(); ++k) delete buckets[k].batch; }
    MixedBatchLCQProblem(const MixedBatchLCQProblem&) = delete;
    MixedBatchLCQProblem& operator=(const MixedBatchLCQProblem&) = delete;

    ReturnValue getReturnValue(int i) const { const BatchLCQProblem* b = batchOf(i); return b ? b->getReturnValue(problems[i].slot) : LCQPOBJECT_NOT_SETUP; }
    AlgorithmStatus getPrimalSolution(int i, double* xOpt) const { return batchOf(i)->getPrimalSolution(problems[i].slot, xOpt); }
    AlgorithmStatus getDualSolution(int i, double* yOpt) const { return batchOf(i)->getDualSolution(problems[i].slot, yOpt); }
    const lcqp_stats_t& getStats(int i) const { return batchOf(i)->getStats(problems[i].slot); }
    int getNumberOfPrimals(int i) const { return problems[i].nV; }
    int getNumberOfDuals(int i) const { return problems[i].nV + problems[i].nC + 2 * problems[i].nComp; }

  private:
    struct Problem {
        int nV, nC, nComp;
        const double *Q, *g, *L, *R, *lbL, *ubL, *lbR, *ubR, *A, *lbA, *ubA, *lb, *ub, *x0, *y0;
        int bucket, slot;
    };
    struct Bucket { int nV, nC, nComp; bool box; std::vector<int> members; BatchLCQProblem* batch; };
    const BatchLCQProblem* batchOf(int i) const
    {
        if (i < 0 || i >= (int)problems.size() || problems[i].bucket < 0 || problems[i].bucket >= (int)buckets.size()) return 0;
        return buckets[problems[i].bucket].batch;
    }
    int device_;
    bool haveOptions;
    Options options;
    std::vector<Problem> problems;
    std::vector<Bucket> buckets;
};

}  // namespace LCQPow
#endif
