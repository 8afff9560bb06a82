import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import os, sys, ctypes, time
os.environ["WJ_PERSIST_STAMPS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from wavjepa_amd import _abi
from wavjepa_amd.data import SyntheticAudioSource
from wavjepa_amd.masking import TimeInverseBlockMasker
from wavjepa_amd.trainer import StepRunner
dev = torch.device("cuda", 0)
model = bench.build_model(dev, 42)
model.trainer.max_steps = 375000
masker = TimeInverseBlockMasker(target_masks_per_context=4, context_mask_prob=0.65, context_mask_length=10, target_prob=0.25, target_length=10, ratio_cutoff=0.1)
src = SyntheticAudioSource(masker, batch_size=32, samples_per_audio=8, n_tokens=model.total_patches, seed=42, n_mask_sets=8, device=dev)
runner = StepRunner(model, gradient_clip_val=5.0)
lib = _abi.load(); lib.wj_debug_persist_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for i in range(60):
    runner.step(src.next_batch(), i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    runner.step(src.next_batch(), 60 + i)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 40 * 1000)
buf = np.zeros(256 * 64, dtype=np.uint64)
lib.wj_debug_persist_stamps(buf.ctypes.data, 256 * 64)
s = buf.reshape(256, 64).astype(np.int64)
clk = s[:, 62:]
last = np.array([s[w, :62][s[w, :62] > 0].max() for w in range(256)])
ghz = (clk[:, 1] - clk[:, 0]) / np.maximum(1, last - s[:, 1]) / 10.0
print("shader clock inside the step's last persistent GEMM: median %.3f GHz (min %.3f max %.3f)" % (np.median(ghz), ghz.min(), ghz.max()))
