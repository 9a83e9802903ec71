"""CPU: the C-ABI call sequence of one EMRT train step (fake ABI, tests/fake_abi.py) with the Python site that made each call.
usage: python tools/r6/call_trace.py [name-filter ...]   (no GPU needed)"""
import collections
import sys
import traceback

sys.path.insert(0, ".")
import torch

from tests import fake_abi
from tests.test_host_logic_cpu import _place

fake = fake_abi.install()
orig = fake.call
sites = []


def call(name, *args):
    st = traceback.extract_stack()[:-1]
    here = [f for f in st if "/emrt_amd/" in f.filename]
    sites.append(" <- ".join("%s:%d %s" % (f.filename.split("/emrt_amd/")[-1], f.lineno, f.name) for f in here[-4:][::-1]))
    orig(name, *args)


fake.call = call
from emrt_amd.src.models.emrt import EMRT
from emrt_amd.src.models.losses import MixSoftmaxCrossEntropyLoss
from emrt_amd.src.models.solver import Momentum, PolynomialDecay
torch.manual_seed(0)
m = _place(EMRT(num_classes=6, backbone="resnet50"))
x, lab = torch.randn(2, 3, 64, 64), torch.randint(0, 6, (2, 64, 64))
m.train()
opt = Momentum(m, PolynomialDecay(0.01, 100), 0.9, 1e-4, 1.0)
m.clear_gradients()
fake.calls.clear()
sites.clear()
out = m(x)
nf = len(fake.calls)
loss = MixSoftmaxCrossEntropyLoss()(out, lab)
loss.backward()
opt.step()
flt = sys.argv[1:]
for i, ((n, a), s) in enumerate(zip(fake.calls, sites)):
    if flt and not any(f in n for f in flt):
        continue
    print("%4d %s %-32s %s" % (i, "F" if i < nf else "B", n, s))
print(collections.Counter(n for n, _ in fake.calls).most_common())
print("calls:", len(fake.calls), "forward:", nf)
