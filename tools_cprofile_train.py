import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import super_resolution_amd as S
dev = torch.device("cuda:0")
net = S.GeneratorRRDB().to(dev).train()
opt = S.FlatAdam(net.parameters(), lr=1e-4)
lr = torch.rand(16, 3, 32, 32, device=dev); hr = torch.rand(16, 3, 128, 128, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.l1_loss(net(lr), hr); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# where the backward's host time goes (it runs on autograd's device thread, out of cProfile's sight)
import time
from super_resolution_amd import ops as _ops, esrgan_model as _em
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
_ops.generator_backward = timed("C generator_backward", _ops.generator_backward)
net.packed_bwd_parameters = timed("packed_bwd_parameters", net.packed_bwd_parameters)
net.packed_parameters = timed("packed_parameters", net.packed_parameters)
_ops.generator_forward_train = timed("C generator_forward_train", _ops.generator_forward_train)
_em._GeneratorFunction.backward = staticmethod(timed("Function.backward", _em._GeneratorFunction.backward))
_em._GeneratorFunction.forward = staticmethod(timed("Function.forward", _em._GeneratorFunction.forward))
for _ in range(5): step()
torch.cuda.synchronize()
for k, v in acc.items():
    print(f"{k:32s} {v / 5 * 1e3:7.2f} ms/step")

# the weight packs alone (they are rebuilt after every optimizer step)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    torch.autograd.graph.increment_version(list(net.parameters()))
    net.packed_parameters(); net.packed_bwd_parameters()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
