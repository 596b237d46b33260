import sys, os, json
sys.argv = ['bench.py']
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
for b, a, dt in ((16, 3, 'bf16'), (48, 1, 'bf16'), (8, 1, 'bf16'), (256, 1, 'bf16')):
    r = bench.secondary_train(dev, b, a, dt, steps=12, warmup=6)
    print(os.environ.get('DX_STEP_GRAPH'), b, a, dt, round(r['ms_per_step'], 3), round(r['value']))
