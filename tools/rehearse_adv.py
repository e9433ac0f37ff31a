#!/usr/bin/env python3
"""Two ranks on one GPU (gloo): one data-parallel adversarial iteration, with a stack dump if it stalls."""
import faulthandler
import os
import socket
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    import torch.distributed as dist
    import super_resolution_amd as S
    from super_resolution_amd.ddp import DistributedGenerator
    faulthandler.dump_traceback_later(60, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")

    def note(m):
        print(f"[rank {rank}] {m}", flush=True)

    net = S.GeneratorRRDB(num_blocks=2).to(dev).train()
    dnet = DistributedGenerator(net, bucket_mb=1.0)
    dnet.broadcast_parameters()
    note("generator broadcast")
    netD = S.Discriminator().to(dev).train()
    vgg = S.VGGFeatureExtractor(pretrained=False).to(dev)
    from super_resolution_amd.ddp import broadcast_tensors
    broadcast_tensors(list(netD.parameters()) + list(netD.buffers()) + list(vgg.parameters()))
    note("D / VGG broadcast")
    optG = S.FlatAdam(net.parameters(), lr=1e-4)
    optD = S.FlatAdam(netD.parameters(), lr=1e-4, reduce_group=dist.group.WORLD)
    lr_b = torch.rand(2, 3, 32, 32, device=dev)
    hr_b = torch.rand(2, 3, 128, 128, device=dev)
    note("pixel step")
    S.pixel_step(dnet, optG, lr_b, hr_b)
    torch.cuda.synchronize()
    note("adversarial step")
    out = S.adversarial_step(dnet, netD, vgg, optG, optD, lr_b, hr_b)
    torch.cuda.synchronize()
    note(f"done {float(out['loss_G']):.4f} {float(out['loss_D']):.4f}")
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
