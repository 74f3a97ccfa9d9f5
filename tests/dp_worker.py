"""Worker of test_two_rank_data_parallel_on_one_gpu: one data-parallel rank (gloo transport, both ranks on cuda:0) -- and, with
ECHR_DP_WORKER_BACKEND=nccl and world size 1, of test_single_rank_rccl_one_call_path (the RCCL code path of every collective on a 1-GPU box)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, out, overlap = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == '1'
    fused_mode = sys.argv[5] in ('fused', 'fused1')          # the one-call path: echr_train_step + hand-over collectives ('fused1': ONE collective)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', port
    backend = os.environ.get('ECHR_DP_WORKER_BACKEND', 'gloo')
    if backend == 'nccl':          # RCCL wants one device per rank: world size 1 on the one-GPU box
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import echr_amd
    from echr_amd import parallel, synth
    from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
    from echr_amd.optim import ClampAdam
    from tests import util as U
    dev = torch.device('cuda', 0)
    # both ranks share cuda:0 here: two 256-workgroup persistent grids must never be half-resident beside each other, so this rehearsal
    # runs the launch-per-phase recurrences -- or, with ECHR_DP_WORKER_COOP=1, persistent grids launched cooperatively (a grid starts only
    # when all of its workgroups can be resident)
    from echr_amd import _lib
    lib = _lib.load()
    if os.environ.get('ECHR_DP_WORKER_COOP') == '1':
        lib.echr_config_set(b'persist_coop', 1)
    else:
        lib.echr_config_set(b'persist', 0)
        lib.echr_config_set(b'persist_bwd', 0)
    opt, params, _ = synth.make_case('c1')
    model = U.build_gpu_model(opt, params, True)
    arena = model.build_arena()
    if overlap:
        parallel.enable_overlap(model)
    optim = ClampAdam(model.parameters(), lr=1e-3, arena=arena)
    crit = LanguageModelCriterion()
    if fused_mode:
        from echr_amd.fused import DataParallelStep, FusedTrainStep
        dp = DataParallelStep(FusedTrainStep(model, optim, grad_clip=0.05), overlap=sys.argv[5] == 'fused', algo=os.environ.get('ECHR_DP_WORKER_ALGO') or None)
    n_early = -1
    for step in range(2):
        vid = synth.make_video(2, 16, 11, opt.CG_vocab_size + 1, seed=500 + 10 * step + rank, T_v=40, video_dim=opt.video_dim,
                               hidden_dim=opt.hidden_dim, lda_dim=opt.video_context_dim)
        tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
        labels = torch.from_numpy(vid['labels'])
        model.set_dropout_state(U.SEED, U.OFFSET + 10 * step + rank)
        if fused_mode:
            # host-side criterion inputs, as bench.py hands them over (active-row compaction inside the call)
            dp(tap, c3d, lda, labels, vid['ind'], vid['soi'], labels[:, 1:].numpy(), vid['masks'][:, 1:])
            n, n_early = dp.n_collectives, dp.n_early
            if step == 0:
                torch.cuda.synchronize()          # (clamp + Adam read the gradient arena, they do not write it)
                grads = {'grad|' + k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
            continue
        optim.zero_grad()
        loss = crit(model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train'), labels[:, 1:].to(dev),
                    torch.from_numpy(vid['masks'])[:, 1:].to(dev))
        loss.backward()
        n = parallel.allreduce_gradients(model)
        if step == 0:
            torch.cuda.synchronize()
            grads = {'grad|' + k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        clip_gradient(optim, 0.05)
        optim.step()
    torch.cuda.synchronize()
    np.savez(out, n_collectives=n, n_early=n_early, **grads, **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
