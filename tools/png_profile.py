"""Where the PNG egress of render_surgery spends its time: GPU only (render + scanlines + device deflate, no fetch), then the
whole ring with host threads.  python tools/png_profile.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor

from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
from omfs_4d_video_gen_amd.engine.io_formats import png_parts_from_zlib_stream as png_from_zlib_stream
from omfs_4d_video_gen_amd.engine.trainer import Renderer, View

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H, N = 1920, 1080, 300000
srig = synthetic.make_rig(0)
rig = FlameRig.from_synthetic(srig)
seq = synthetic.make_flame_sequence(n, 0)
cams = synthetic.make_camera_arc(W, H, 16)
rr = Renderer(rig, seq, synthetic.make_gaussians(N, rig.n_faces, 0), W, H)
frames = [View(cams[i % 16], i) for i in range(n)]


def timed(fn, label):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{label}: {dt / n * 1e3:.3f} ms per frame ({n / dt:.0f} fps)")


for v in frames[:8]:
    rr.render_png_stream(v, 32)
timed(lambda: [rr.render(v, rgb8=True) for v in frames], "render + rgb8")
timed(lambda: [(rr.render(v), rr.rast.to_png_rows()) for v in frames], "render + scanlines")
timed(lambda: [rr.render_png_stream(v, 32) for v in frames], "render + scanlines + device deflate (no fetch)")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
rr.render(frames[0]); rr.rast.to_png_rows(); torch.cuda.synchronize()
ev[0].record(); rr.rast.to_png_stream(); ev[1].record(); torch.cuda.synchronize()
print(f"deflate + assemble kernels: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us, stream {int(rr.rast._png_len.item())} bytes")
for workers in (4, 16):
    def loop():
        with ThreadPoolExecutor(max_workers=workers) as pool:
            futs = []
            for v in frames:
                if len(futs) >= 32:
                    futs.pop(0).result()
                k, e = rr.render_png_stream(v, 32)
                futs.append(pool.submit(lambda k=k, e=e: len(png_from_zlib_stream(rr.fetch_png_stream(k, e), W, H))))
            for f in futs:
                f.result()
    timed(loop, f"whole ring, {workers} host threads")
t = time.perf_counter()
k, e = rr.render_png_stream(frames[0], 32)
mv = rr.fetch_png_stream(k, e)
t1 = time.perf_counter()
for _ in range(20):
    png_from_zlib_stream(mv, W, H)
print(f"host: fetch {1e3 * (t1 - t):.2f} ms, PNG framing + CRC {(time.perf_counter() - t1) / 20 * 1e3:.2f} ms per frame ({len(mv)} bytes)")
