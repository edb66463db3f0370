import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-ncc", "--no-cpu-baseline", "--no-stages"], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1]); p = d["roofline"]["pass_ms"]
print(f"{d['ms_per_step']:.2f} ms  y {p['y_forward']:.2f}/{p['y_inverse']:.2f} z {p['z_conv']:.3f} x {p['x_fused_ratio']:.2f}/{p['x_fused_update']:.2f}")
