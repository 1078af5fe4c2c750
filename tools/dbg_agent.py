import sys; sys.path.insert(0,'/root/repo')
import tests.test_gpu_agent as T
import torch
from tests.test_gpu_net import rel_rms, cosine
# monkeypatch: run the test body but print all grads
import numpy as np
src = open('/root/repo/tests/test_gpu_agent.py').read()
src = src.replace("    assert not bad, bad[:10]", "    rows=[(n, rel_rms(g.cpu(), gref[n].grad), cosine(g.cpu(), gref[n].grad)) for n,g in agent.named_grads() if n.endswith('weight')]\n    [print('%-34s rel %.3f cos %.4f'%r) for r in rows]")
from tests import bf16_emul as BE
BE.ROUND_GRADS = True
exec(compile(src, 'x', 'exec'))
test_agent_logits_loss_and_gradients_local()
