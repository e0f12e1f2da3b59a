import sys; sys.path.insert(0, '.')
from pegainfer_amd.qwen3 import QWEN3_4B, Qwen3Engine
eng = Qwen3Engine(QWEN3_4B, num_kv_pages=64, max_batch_size=1).fill_synthetic()
for which, name, mb in [(0,'qkv',31.46),(1,'o',20.97),(2,'gate_up',99.6),(3,'down',49.8)]:
    cold = eng.bench_gemv(which, 360); warm = eng.bench_gemv(which+10, 360)
    print(f"{name}: cold {cold*1e3:.2f} us ({mb/cold/1e3:.0f} GB/s)  same-layer {warm*1e3:.2f} us ({mb/warm/1e3:.0f} GB/s)")
