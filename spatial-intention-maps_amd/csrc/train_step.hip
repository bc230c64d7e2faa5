// libsimq: the whole TD step of train.py:108-141 as one library call (simq_train_step) -- three forwards, TD target + Huber loss,
// backward (with the two gradient all-reduces of the data-parallel form), clip + SGD, weight-cache refresh.
#include "plan.h"

using namespace simq;

namespace {
// out4 -> pinned host memory without a stream synchronisation: the plan's copy stream + two events of the current device (PlanStreams)
// copy_on: a stream of the CALLER's to run the copy on instead of the plan's copy stream -- simq_train_step hands over the third stream when
// the caller named one: it is idle from the end of the forward phase to the next step, and the caller has TESTED that it does not share the
// launch stream's hardware queue (simq_train_args.third_stream), which nobody can say of a stream the library creates
int loss_copy(const simq_plan* plan, const float* d_out4, float* h_out4, hipStream_t producer, bool own_stream, hipStream_t copy_on = nullptr) {
    PlanStreams* c = nullptr;
    int dev = 0;
    RC(plan_streams(plan, &c, &dev));
    SIMQ_REQUIRE(c, "train_step: device index %d out of range", dev);
    if (producer) {                         // the copy stream / events are created on the CURRENT device: it must be the producer stream's
        hipDevice_t sdev = 0;
        SIMQ_CHECK_HIP(hipStreamGetDevice(producer, &sdev));
        SIMQ_REQUIRE((int)sdev == dev, "train_step: the stream belongs to device %d, the calling thread's current device is %d", (int)sdev, dev);
    }
    if (!c->copy_ready) {
        std::lock_guard<std::mutex> lk(plan->mu);
        SIMQ_CHECK_HIP(hipEventCreateWithFlags(&c->copy_ready, hipEventDisableTiming));
        SIMQ_CHECK_HIP(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    }
    if (own_stream && !copy_on && !c->copy) {
        std::lock_guard<std::mutex> lk(plan->mu);
        SIMQ_CHECK_HIP(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    }
    hipStream_t s = producer;
    if (own_stream) {                       // the copy must not queue behind the backward pass that follows on `producer`
        s = copy_on ? copy_on : c->copy;
        SIMQ_CHECK_HIP(hipEventRecord(c->copy_ready, producer));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(s, c->copy_ready, 0));
    }
    SIMQ_CHECK_HIP(hipMemcpyAsync(h_out4, d_out4, 4 * sizeof(float), hipMemcpyDeviceToHost, s));
    SIMQ_CHECK_HIP(hipEventRecord(c->copy_done, s));
    c->copy_pending = true;
    return 0;
}
}  // namespace

extern "C" {

int simq_train_loss_wait(const simq_plan* plan) {
    SIMQ_REQUIRE(plan, "train_loss_wait: NULL plan");
    PlanStreams* c = nullptr;
    int dev = 0;
    RC(plan_streams(plan, &c, &dev));
    SIMQ_REQUIRE(c && c->copy_pending, "train_loss_wait: no simq_train_step with loss_host on this plan and device (%d)", dev);
    SIMQ_CHECK_HIP(hipEventSynchronize(c->copy_done));
    c->copy_pending = false;
    return 0;
}

int simq_train_step(const simq_train_args* a) {
    SIMQ_REQUIRE(a && a->plan, "train_step: NULL argument");
    SIMQ_REQUIRE(a->struct_bytes == (int)sizeof(simq_train_args), "train_step: simq_train_args.struct_bytes = %d, this library's struct has %d bytes",
                 a->struct_bytes, (int)sizeof(simq_train_args));
    SIMQ_REQUIRE(a->params && a->wcache && a->bnbuf && a->grads && a->momentum_buf && a->ws_train && a->ws_tmp && a->t_params &&
                 a->t_wcache && a->t_bnbuf && a->t_ws && a->state && a->next_state && a->action && a->reward && a->nonfinal_pos &&
                 a->q && a->q_tgt && a->nsv && a->vals && a->q_sa && a->y && a->td && a->out4 && a->opt_scratch,
                 "train_step: NULL buffer");
    SIMQ_REQUIRE(!a->use_double_dqn || a->num_nonfinal == 0 || (a->q_next && a->best), "train_step: double DQN needs q_next and best");
    SIMQ_REQUIRE(!(a->comm && a->sync_bn) || a->global_nonfinal >= a->num_nonfinal, "train_step: sync_bn needs global_nonfinal (>= num_nonfinal)");
    // single process: the reference itself fails on a minibatch without any non-final next state (torch.cat([]) at train.py:112);
    // a data-parallel SHARD may have none and still has to join the collectives
    SIMQ_REQUIRE(a->batch >= 1 && a->num_nonfinal >= (a->comm ? 0 : 1) && a->num_nonfinal <= a->batch && a->global_batch >= a->batch,
                 "train_step: batch=%d num_nonfinal=%d global_batch=%d", a->batch, a->num_nonfinal, a->global_batch);
    const simq_plan* p = a->plan;
    hipStream_t main = static_cast<hipStream_t>(a->stream), side = static_cast<hipStream_t>(a->side_stream);
    const int n = p->cout * 96 * 96, B = a->batch, Nn = a->num_nonfinal;
    // fork / join events of the caller's side stream: the plan's, per device (events belong to the device they were created on)
    PlanStreams* ps = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_wfork = nullptr, ev_wjoin = nullptr, ev_wdone0 = nullptr, ev_wdone1 = nullptr;
    if (side) {
        int dev = 0;
        RC(plan_streams(p, &ps, &dev));
        SIMQ_REQUIRE(ps, "train_step: device index %d out of range", dev);
        if (!ps->step_ev[0]) {
            std::lock_guard<std::mutex> lk(p->mu);
            for (int i = 0; i < 7; ++i) SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ps->step_ev[i], hipEventDisableTiming));
        }
        ev_fork = ps->step_ev[0]; ev_join = ps->step_ev[1]; ev_wfork = ps->step_ev[2]; ev_wjoin = ps->step_ev[3];
        ev_wdone0 = ps->step_ev[4]; ev_wdone1 = ps->step_ev[5];
    }
    const int fwd_overlap = p->opt.fwd_overlap;      // simq_plan_options.fwd_overlap (2: three forwards side by side)
    const int late_block = p->opt.early_target_after_block;
    // SyncBN option of the data-parallel form: the train-mode BatchNorms see the statistics of the global minibatch
    simq_sync sync_storage{comm_reduce_f64, a->comm, a->global_batch, a->comm ? simq_comm_world_size(a->comm) : 1};
    const simq_sync* sync = (a->comm && a->sync_bn) ? &sync_storage : nullptr;
    // Three forwards side by side (round 4).  The policy's no-grad forward over the next states (train.py:121) reads nothing the grad-mode
    // forward (train.py:114) writes except the BatchNorm running statistics, which BOTH update (the policy net is in train mode) and no
    // forward reads: it runs on a third stream from the start of the step with that update deferred -- its committing blocks leave
    // [mean | unbiased variance] in fp64 (BnRef::defer) and one launch applies them behind the grad-mode forward's update, the same fp64
    // expression on the same values in the reference's order: the buffers are bit-identical to the serial order's.  The transform-domain
    // forwards alternate HBM-bound transforms and matrix-bound GEMMs; side by side the three fill each other's phases
    // (fp32 configs[1] +4.7 ... +5.9 %, bf16 configs[2] +2.6 %).  Not under SyncBN (its collectives order the streams); the plain
    // data-parallel step has no collective before its backward pass and takes it.
    const bool three = fwd_overlap == 2 && side && Nn > 0 && a->use_double_dqn && !sync;
    // target_stream is an ordering promise of the caller (it made that stream wait for the target forward's inputs INSTEAD of this step's
    // launch stream): only the three-forward form keeps it -- any other form would silently run that forward behind the launch stream
    SIMQ_REQUIRE(!a->target_stream || three, "train_step: target_stream needs the three-forward form (fwd_overlap = 2, a side stream, double DQN, "
                 "non-final next states, no SyncBN)");
    if (three) {
        if (!ps->third_ev) {
            std::lock_guard<std::mutex> lk(p->mu);
            SIMQ_CHECK_HIP(hipEventCreateWithFlags(&ps->third_ev, hipEventDisableTiming));
        }
        if (!a->third_stream && !ps->third) {
            std::lock_guard<std::mutex> lk(p->mu);
            SIMQ_CHECK_HIP(hipStreamCreateWithFlags(&ps->third, hipStreamNonBlocking));
        }
        // (the caller's stream when it names one: it has chosen streams that do not share a hardware queue, see simq_train_args.third_stream)
        hipStream_t third = a->third_stream ? static_cast<hipStream_t>(a->third_stream) : ps->third;
        // (target_stream: the caller ordered it behind the target forward's inputs -- no wait for this step's or the previous step's work)
        hipStream_t tstream = a->target_stream ? static_cast<hipStream_t>(a->target_stream) : side;
        // ... except for a point of the PREVIOUS step's backward walk (simq_plan_options.early_target_after_block): started at once, this forward
        // races through the start of that backward pass, where the dgrads and the piped weight gradients already fill the matrix cores, and
        // is done before the walk reaches the narrow layers, the optimiser step and the weight-cache refresh, which then run alone.  Held back
        // until the walk is in front of block 4, it covers those instead (alternating A/B on one box, tools/ab_step.py, 80 steps: fp32
        // configs[1] 3934-3950 -> 4006-4015 tr/s, bf16 configs[2] 14 311-14 471 -> 14 567-14 765; profiles/r05_ab_early_target_delay.txt)
        if (late_block >= 0 && a->target_stream && ps->late_recorded) SIMQ_CHECK_HIP(hipStreamWaitEvent(tstream, ps->step_ev[6], 0));
        SIMQ_CHECK_HIP(hipEventRecord(ev_fork, main));
        if (!a->target_stream) SIMQ_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(third, ev_fork, 0));
        RC(simq_forward(p, SIMQ_MODE_EVAL, Nn, a->t_params, a->t_wcache, a->t_bnbuf, a->next_state, a->q_tgt, a->t_ws, tstream));
        SIMQ_CHECK_HIP(hipEventRecord(ev_join, tstream));
        Ctx cn{p, Nn, a->params, nullptr, a->bnbuf, static_cast<char*>(a->ws_tmp), make_layout(p, Nn), third};
        cn.wc = static_cast<char*>(const_cast<void*>(a->wcache)); cn.W = make_wlayout(p);
        cn.defer_running = true;
        RC(forward_impl(cn, SIMQ_MODE_TRAIN_NOGRAD, a->next_state, a->q_next));
        RC(launch_q_argmax(a->q_next, Nn, n, a->best, nullptr, third));
        SIMQ_CHECK_HIP(hipEventRecord(ps->third_ev, third));
        RC(forward_sync_inplace(p, B, a->params, a->wcache, a->bnbuf, a->state, a->q, a->ws_train, main, sync));   // train.py:114 (a->state in place)
        SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ps->third_ev, 0));
        RC(launch_bn_running_deferred(a->bnbuf, reinterpret_cast<const double*>(cn.ws + cn.L.defer), p->nbnbuf, main));     // update #2
        SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ev_join, 0));
        RC(launch_q_gather(a->q_tgt, Nn, n, a->best, a->vals, main));
    } else {
    if (fwd_overlap == 1 && side) {                      // (A-B: the target-net forward forked at the start of the step)
        SIMQ_CHECK_HIP(hipEventRecord(ev_fork, main));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    }
    RC(forward_sync_inplace(p, B, a->params, a->wcache, a->bnbuf, a->state, a->q, a->ws_train, main, sync));   // train.py:114 (a->state in place)
    // the target-net forward depends on nothing the policy net computes: side stream, joined before its Q-map is read.  It is
    // forked BEHIND the policy's train-mode forward so that it overlaps the policy's next-state forward: both run on the
    // ~29 non-final samples, whose tiles do not fill whole rounds of the CUs, and fill each other's tails (+1.7 % on the step
    // over starting it beside the perfectly tiled 32-sample forward)
    if (side && fwd_overlap != 1) {
        SIMQ_CHECK_HIP(hipEventRecord(ev_fork, main));
        SIMQ_CHECK_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    }
    if (Nn > 0) {
    RC(simq_forward(p, SIMQ_MODE_EVAL, Nn, a->t_params, a->t_wcache, a->t_bnbuf, a->next_state, a->q_tgt, a->t_ws, side ? side : main));
    if (side) SIMQ_CHECK_HIP(hipEventRecord(ev_join, side));
    if (a->use_double_dqn) {                                                                                          // train.py:119-122
        // (under SyncBN this forward normalises over the non-final next states of ALL ranks)
        simq_sync sync_nf = sync_storage;
        sync_nf.global_batch = a->global_nonfinal;
        RC(simq_forward_sync(p, SIMQ_MODE_TRAIN_NOGRAD, Nn, a->params, a->wcache, a->bnbuf, a->next_state, a->q_next, a->ws_tmp, main,
                             sync ? &sync_nf : nullptr));
        RC(launch_q_argmax(a->q_next, Nn, n, a->best, nullptr, main));
        if (side) SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ev_join, 0));
        RC(launch_q_gather(a->q_tgt, Nn, n, a->best, a->vals, main));
    } else {                                                                                                          // train.py:124
        if (side) SIMQ_CHECK_HIP(hipStreamWaitEvent(main, ev_join, 0));
        RC(launch_q_argmax(a->q_tgt, Nn, n, nullptr, a->vals, main));
    }
    }
    }
    if (Nn == 0 && sync && a->use_double_dqn && a->global_nonfinal > 0) {     // all-terminal shard: zeros into the other ranks' reductions
        simq_sync sync_nf = sync_storage;
        sync_nf.global_batch = a->global_nonfinal;
        RC(simq_forward_sync_null(p, 1, a->bnbuf, a->ws_tmp, main, &sync_nf));
    }
    RC(launch_scatter_next_values(a->vals, a->nonfinal_pos, Nn, a->nsv, B, main));                                    // train.py:116-122
    RC(launch_td_huber(a->q, B, n, a->action, a->reward, a->nsv, a->gamma, 1.0f / (float)a->global_batch, a->q_sa, a->y, a->td,
                       a->out4, a->dq, main));                                                                       // train.py:115,126-129
    if (a->loss_host && !a->comm)                                                          // train.py:137-139: the loss is final here
        RC(loss_copy(p, a->out4, a->loss_host, main, true, (three && a->third_stream) ? static_cast<hipStream_t>(a->third_stream) : nullptr));
    if (a->comm) {
        // data parallel: the four loss sums are summed over the ranks HERE, not behind the last gradient bucket, and copied to the host on the
        // communicator's stream right behind that all-reduce -- a caller that waits for the loss (simq_train_loss_wait) gets it when the forwards
        // of all ranks are done, as in the single-GPU form, and enqueues its next step while this one's backward pass runs.  Measured with a
        // 1-rank communicator on one GPU (profiles/r06_ab_early_target_delay_dp.txt): with the loss behind bucket 2 the host stood still for the
        // whole backward pass and the data-parallel form was 6.5 % slower than the plain step.
        RC(comm_allreduce(a->comm, a->out4, 4, SIMQ_COMM_F32, main));
        if (a->loss_host) RC(loss_copy(p, a->out4, a->loss_host, comm_stream(a->comm), false));
    }
    const float gscale = 1.0f / (float)a->global_batch;
    auto backward = [&](int phase) {                                                                                 // train.py:131-132
        if (int rc = check_sync(sync, B)) return rc;
        return backward_sync_side(p, B, a->params, a->wcache, a->dq, a->action, a->q_sa, a->y, gscale, a->grads, a->ws_train, phase, main, sync,
                                  side, ev_wfork, ev_wjoin, ev_wdone0, ev_wdone1, a->state,
                                  (ps && late_block >= 0) ? ps->step_ev[6] : nullptr, late_block);
    };
    if (!a->comm) {
        RC(backward(0));
    } else {
        // data parallel (DataParallel's reduce-add, policies.py:39, as RCCL all-reduces): the head + layer4 bucket (75 % of the
        // bytes) is final after phase 1 and travels on the communicator's stream while phase 2 differentiates layers 3..1 + stem
        const int64_t split = simq_grad_bucket_split(p);
        RC(backward(1));
        RC(comm_allreduce(a->comm, a->grads + split, p->nparams - split, SIMQ_COMM_F32, main));
        RC(backward(2));
        RC(comm_allreduce(a->comm, a->grads, split, SIMQ_COMM_F32, main));
        RC(comm_wait(a->comm, main));
    }
    // (both phases of the data-parallel form together visit every block once.)  A step that recorded no such event -- early_target_after_block < 0 --
    // clears the mark: a later step must not wait on an event of some unrelated earlier walk
    if (ps) ps->late_recorded = late_block >= 0;
    RC(launch_clip_sgd(a->params, a->grads, a->momentum_buf, p->nparams, a->max_norm, a->lr, a->momentum, a->weight_decay,
                       a->first_step, a->opt_scratch, a->total_norm, main));                                         // train.py:133-135
    return simq_weights_prepare(p, a->params, a->wcache, main);
}

int64_t simq_grad_bucket_split(const simq_plan* plan) { return plan ? plan->blocks[kPhaseSplitBlock].c1.w_off : -1; }

}  // extern "C"
