"""Host driver of the splat kernels: project -> bin/sort -> composite (+ backward, loss, Adam).

All buffers are allocated once per (N, width, height) and reused; every call only enqueues
kernels on the current HIP stream (no host sync), so a frame or a training step can be captured
into a hipGraph by the caller.
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np
import torch

from .. import _lib as L
from .gaussians import GaussianModel, NPLANES

TILE = 16


def make_camera_struct(cam: dict, sh_degree: int = 3, bg=(0.0, 0.0, 0.0)) -> L.CameraC:
    """cam: dict from synthetic.make_camera / dataset reader (world_to_view 4x4, fl_x, fl_y, tanfov*, size)."""
    c = L.CameraC()
    w2v = np.asarray(cam["world_to_view"], np.float32)
    c.view[:] = [float(v) for v in w2v[:3, :4].reshape(-1)]
    c.cam_pos[:] = [float(v) for v in np.asarray(cam["cam_pos"], np.float32)]
    c.fx, c.fy = float(np.float32(cam["fl_x"])), float(np.float32(cam["fl_y"]))
    c.width, c.height = int(cam["width"]), int(cam["height"])
    c.cx, c.cy = (c.width - 1) * 0.5, (c.height - 1) * 0.5
    c.limx = float(np.float32(1.3) * np.float32(cam["tanfovx"]))
    c.limy = float(np.float32(1.3) * np.float32(cam["tanfovy"]))
    c.sh_degree = int(sh_degree)
    c.bg[:] = [float(b) for b in bg]
    return c


class Rasterizer:
    def __init__(self, n: int, width: int, height: int, device="cuda", dup_capacity: int | None = None,
                 sort_lds_pairs: int = 0, n_capacity: int | None = None):
        """n_capacity: per-Gaussian buffers are sized for this many Gaussians (densification grows N)."""
        self.device = torch.device(device)
        self.n_capacity = int(n_capacity if n_capacity else n)
        if n > self.n_capacity:
            raise ValueError("n exceeds n_capacity")
        n = self.n_capacity
        self.n, self.width, self.height = int(n), int(width), int(height)
        self.gx, self.gy = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
        self.n_tiles = self.gx * self.gy
        # Default pair capacity: 20 pairs per Gaussian at 1080p, growing with the image area beyond that (a splat covers four times
        # as many tiles at 3840x2160).  Measured on the synthetic heads (tools/early_d.py, profiles/r05_early_tile_pairs.json): the
        # UNTRAINED cloud is the heaviest, 15.8 pairs per Gaussian at 1080p for 300 k and 500 k Gaussians, 12.1 per unit of area at
        # 4K, 5.8 at 512 x 512; a trained cloud holds 11-12.  The capacity sizes keys / keys_tmp / sorted_ids, the checkpoint
        # buffer (4 KB per 128 pairs) and the grid of the backward pass (4 waves per 128 pairs + 4 per tile), so it is kept close
        # (a quarter above the heaviest cloud measured: 16 would leave 1.4 %, and a renderer has no way to redo a frame inside
        # render()): rounds 1-4 allocated 48 per Gaussian, three quarters of it never touched.  Exceeding it is flagged by the device and every
        # caller grows the buffers and redoes the work: engine/train.py (rollback to the last clean snapshot), engine/render.py
        # (the split is rendered again), bench.py (the warm-up is repeated), Trainer users through rast.overflowed().
        area = max(1.0, (width * height) / float(1920 * 1080))
        self.dup_capacity = int(dup_capacity if dup_capacity else max(1 << 20, int(20 * n * area)))
        dev = self.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        # projected-splat records: three planar arrays, or -- a library built with -DOMFS_REC_STRIDE=4 (A/B builds; 0.3 % slower) --
        # ONE 64-byte record per Gaussian of which g0 / g1 / g2 are views
        if L.load().omfs_record_stride() == 4:
            self.rec = z(n, 16)
            self.g0, self.g1, self.g2 = self.rec[:, 0:4], self.rec[:, 4:8], self.rec[:, 8:12]
        else:
            self.rec = None
            self.g0, self.g1, self.g2 = z(n, 4), z(n, 4), z(n, 4)
        self.tile_count = z(self.n_tiles, dt=torch.int32)
        self.tile_start = z(self.n_tiles + 1, dt=torch.int32)
        self.tile_cursor = z(self.n_tiles, dt=torch.int32)
        self.tile_order = z(self.n_tiles, dt=torch.int32)
        self.keys = z(self.dup_capacity, 2, dt=torch.int32)
        self.keys_tmp = z(self.dup_capacity, 2, dt=torch.int32)
        self.sorted_ids = z(self.dup_capacity, dt=torch.int32)
        self.status = z(2, dt=torch.int32)      # word 0: overflow flag; word 1: the library's tile-test replay stamp
        self.seg_capacity = self.n_tiles + self.dup_capacity // L.SEG + 1
        self.seg_ckpt = z(self.seg_capacity, 256, 4)
        self.order_seg0 = z(self.n_tiles + 1, dt=torch.int32)
        self.image = z(3, height, width)
        self.final_T = z(height, width)
        self.n_contrib = z(height, width, dt=torch.int32)
        self.n_visible = z(1, dt=torch.int32)
        self.rb = L.RasterBuffersC(L.ptr(self.g0), L.ptr(self.g1), L.ptr(self.g2), L.ptr(self.tile_count),
                                   L.ptr(self.tile_start), L.ptr(self.tile_cursor), L.ptr(self.tile_order),
                                   L.ptr(self.keys), L.ptr(self.keys_tmp), L.ptr(self.sorted_ids),
                                   self.dup_capacity, int(sort_lds_pairs or os.environ.get("OMFS_SORT_LDS_PAIRS", 0)), L.ptr(self.status),
                                   L.ptr(self.seg_ckpt), L.ptr(self.order_seg0), self.seg_capacity, L.ptr(self.image),
                                   L.ptr(self.final_T), L.ptr(self.n_contrib), 0, 0, 0)
        # backward-side buffers are created on first use
        self.dsplat = None
        self.dsplat_fx = None
        self.deterministic = os.environ.get("OMFS_DETERMINISTIC", "0") == "1"
        self.dimage = None
        self.loss = None
        self.loss_scratch = None
        self.rgb8 = None

    # ------------------------------------------------------------------ forward
    def _gauss(self, model: GaussianModel) -> L.GaussiansC:
        if model.n > self.n_capacity:
            raise ValueError(f"rasterizer was sized for {self.n_capacity} Gaussians, model has {model.n}")
        self.n = model.n
        return L.GaussiansC(model.n, model.n_pad, L.ptr(model.params), L.ptr(model.binding))

    def project(self, model: GaussianModel, face_xf: torch.Tensor, cam: L.CameraC):
        if cam.width != self.width or cam.height != self.height:
            raise ValueError("camera size does not match the rasterizer's buffers")
        if face_xf.dtype != torch.float32 or not face_xf.is_contiguous() or face_xf.shape[-1] != 16:
            raise ValueError("face_xf must be a contiguous float32 [F][16] tensor")
        lib = L.load()
        L.check(lib.omfs_project_fwd(self._gauss(model), L.ptr(face_xf), cam, self.rb, L.stream_ptr()), "omfs_project_fwd")

    def bin_sort(self, model: GaussianModel, cam: L.CameraC):
        L.check(L.load().omfs_bin_sort(self._gauss(model), cam, self.rb, L.stream_ptr()), "omfs_bin_sort")

    def composite(self, cam: L.CameraC):
        L.check(L.load().omfs_composite_fwd(cam, self.rb, L.stream_ptr()), "omfs_composite_fwd")

    def forward(self, model: GaussianModel, face_xf: torch.Tensor, cam: L.CameraC) -> torch.Tensor:
        """Enqueue one view; returns the (reused) image tensor [3][H][W]."""
        self.project(model, face_xf, cam)
        self.bin_sort(model, cam)
        self.composite(cam)
        return self.image

    def to_rgb8(self) -> torch.Tensor:
        if self.rgb8 is None:
            self.rgb8 = torch.zeros(self.height, self.width, 3, dtype=torch.uint8, device=self.device)
        L.check(L.load().omfs_image_to_rgb8(L.ptr(self.image), self.width, self.height, L.ptr(self.rgb8), L.stream_ptr()),
                "omfs_image_to_rgb8")
        return self.rgb8

    def to_png_stream(self, stream: torch.Tensor | None = None, length: torch.Tensor | None = None):
        """The frame as a complete zlib stream of its PNG scanlines, deflated ON THE DEVICE (omfs_png_deflate): returns
        (stream uint8 [capacity], length int32 [1]) device tensors -- the first `length` bytes are the payload of the PNG's IDAT
        chunk (io_formats.png_from_zlib_stream).  `stream` / `length`: caller-owned outputs (a ring of frames in flight)."""
        rows = self.to_png_rows()
        lib = L.load()
        if getattr(self, "_png_scratch", None) is None:
            stride = int(lib.omfs_png_slot_stride(self.width))
            z = lambda n, dt: torch.zeros(n, dtype=dt, device=self.device)
            self._png_scratch = (z(self.height * stride, torch.uint8), z(self.height, torch.int32), z(2 * self.height, torch.int32))
            self.png_stream_capacity = self.height * stride + 16
            self._png_stream, self._png_len = z(self.png_stream_capacity, torch.uint8), z(1, torch.int32)
        slots, sizes, adler = self._png_scratch
        stream = self._png_stream if stream is None else stream
        length = self._png_len if length is None else length
        if stream.numel() < self.png_stream_capacity or stream.dtype != torch.uint8 or not stream.is_contiguous():
            raise ValueError(f"stream must be a contiguous uint8 tensor of at least {self.png_stream_capacity} bytes")
        L.check(lib.omfs_png_deflate(L.ptr(rows), self.width, self.height, L.ptr(slots), L.ptr(sizes), L.ptr(adler), L.ptr(stream),
                                     int(stream.numel()), L.ptr(length), L.stream_ptr()), "omfs_png_deflate")
        return stream, length

    def to_png_rows(self) -> torch.Tensor:
        """[H][1 + 3W] uint8: the frame as PNG scanlines (filter byte 0 per row), ready for io_formats.encode_png_rows."""
        if getattr(self, "png_rows", None) is None:
            # (+16 bytes: the deflate kernel reads the aligned words that cover a scanline)
            self._png_rows_store = torch.zeros(self.height * (1 + 3 * self.width) + 16, dtype=torch.uint8, device=self.device)
            self.png_rows = self._png_rows_store[:self.height * (1 + 3 * self.width)].view(self.height, 1 + 3 * self.width)
        L.check(L.load().omfs_image_to_png_rows(L.ptr(self.image), self.width, self.height, L.ptr(self.png_rows), L.stream_ptr()),
                "omfs_image_to_png_rows")
        return self.png_rows

    def overflowed(self) -> bool:
        """Host sync: did the device flag a tile-list capacity overflow since the flag was last cleared?"""
        return bool(int(self.status[0].item()) & 1)

    def check_status(self):
        """Host sync: raise if the device flagged a capacity overflow."""
        if self.overflowed():
            raise L.OmfsError(f"tile-list capacity {self.dup_capacity} exceeded; re-create the Rasterizer with a larger dup_capacity")

    def grow_dup_capacity(self, factor: float = 2.0) -> int:
        """Re-allocate the pair-sized buffers (keys, keys_tmp, sorted_ids, checkpoints) `factor` times larger and clear the
        overflow flag.  An iteration that overflowed rendered empty lists (the scan zeroes them), i.e. contributed next to
        no gradient; the caller simply goes on."""
        torch.cuda.synchronize(self.device)
        self.dup_capacity = int(self.dup_capacity * factor)
        dev = self.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        self.keys = z(self.dup_capacity, 2, dt=torch.int32)
        self.keys_tmp = z(self.dup_capacity, 2, dt=torch.int32)
        self.sorted_ids = z(self.dup_capacity, dt=torch.int32)
        self.seg_capacity = self.n_tiles + self.dup_capacity // L.SEG + 1
        self.seg_ckpt = z(self.seg_capacity, 256, 4)
        self.status.zero_()
        rb = self.rb
        rb.keys, rb.keys_tmp, rb.sorted_ids = L.ptr(self.keys), L.ptr(self.keys_tmp), L.ptr(self.sorted_ids)
        rb.dup_capacity, rb.seg_ckpt, rb.seg_capacity = self.dup_capacity, L.ptr(self.seg_ckpt), self.seg_capacity
        return self.dup_capacity

    # ------------------------------------------------------------------ backward
    def _ensure_bwd(self):
        if self.dsplat is None:
            dev = self.device
            self.dsplat = torch.zeros(self.n_capacity, 16, device=dev)
            # OMFS_DETERMINISTIC=1: 64-bit fixed-point accumulators beside the float records (omfs_grad_buffers.dsplat_fx): the
            # gradient sums no longer depend on the order the waves arrive in, two runs of the same training are bit-identical
            self.dsplat_fx = torch.zeros(self.n_capacity, 16, dtype=torch.int64, device=dev) if self.deterministic else None
            self.dimage = torch.zeros(3, self.height, self.width, device=dev)
            self.loss = torch.zeros(1, device=dev)
            self.loss_scratch = torch.zeros(3 * 3 * self.height * self.width + L.LOSS_TAIL, device=dev)    # three maps + one loss partial per strip

    def loss_l1_ssim(self, target: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
        """Writes this view's loss into self.loss (device scalar) and dL/dimage into self.dimage."""
        self._ensure_bwd()
        if target.shape != self.image.shape or target.dtype != torch.float32 or not target.is_contiguous():
            raise ValueError("target must be a contiguous float32 [3][H][W] tensor")
        L.check(L.load().omfs_loss_l1_ssim(L.ptr(self.image), L.ptr(target), self.width, self.height, float(lambda_dssim),
                                           L.ptr(self.dimage), L.ptr(self.loss), L.ptr(self.loss_scratch), L.stream_ptr()),
                "omfs_loss_l1_ssim")
        return self.loss

    def grad_buffers(self, grads, dimage, densify_stats=None, dface=None, drgb_out=None, dir_out=None) -> "L.GradBuffersC":
        """omfs_grad_buffers of this rasteriser's backward pass (with the fixed-point accumulators in deterministic mode)."""
        self._ensure_bwd()
        return L.GradBuffersC(L.ptr(self.dsplat), L.ptr(grads), L.ptr(dimage), L.ptr(densify_stats), L.ptr(dface), L.ptr(drgb_out),
                              L.ptr(dir_out), L.ptr(self.dsplat_fx), self.n_capacity if self.dsplat_fx is not None else 0)

    def backward(self, model: GaussianModel, face_xf: torch.Tensor, cam: L.CameraC, grads: torch.Tensor,
                 dimage: torch.Tensor | None = None, reg=(0.01, 1.0, 1.0, 0.6), dface: torch.Tensor | None = None):
        """dL/dimage (default: self.dimage from loss_l1_ssim) -> grads [59][n_pad] (overwritten);
        dface [F][16] (optional, zeroed by the caller) += dL/d(triangle frame records)."""
        self._ensure_bwd()
        lib = L.load()
        s = L.stream_ptr()
        dimg = self.dimage if dimage is None else dimage
        if dimg.shape != self.image.shape or dimg.dtype != torch.float32 or not dimg.is_contiguous():
            raise ValueError("dimage must be a contiguous float32 [3][H][W] tensor")
        if grads.shape != (NPLANES, model.n_pad) or grads.dtype != torch.float32 or not grads.is_contiguous():
            raise ValueError("grads must be a contiguous float32 [59][n_pad] tensor")
        self.dsplat.zero_()
        gb = self.grad_buffers(grads, dimg, dface=dface)
        L.check(lib.omfs_composite_bwd(cam, self.rb, gb, s), "omfs_composite_bwd")
        L.check(lib.omfs_count_visible(self.rb, self.n, L.ptr(self.n_visible), s), "omfs_count_visible")
        rp = L.RegParamsC(float(reg[0]), float(reg[1]), float(reg[2]), float(reg[3]), L.ptr(self.n_visible))
        L.check(lib.omfs_project_bwd(self._gauss(model), L.ptr(face_xf), cam, self.rb, gb, rp, s), "omfs_project_bwd")


class Adam:
    """Fused per-Gaussian Adam over the [59][n_pad] SoA (torch.optim.Adam semantics)."""

    def __init__(self, model: GaussianModel, lr_planes, beta1=0.9, beta2=0.999, eps=1e-15):
        self.model = model
        self.m = torch.zeros_like(model.params)
        self.v = torch.zeros_like(model.params)
        self.step_count = 0
        self.ap = L.AdamParamsC()
        self.set_lr(lr_planes)
        self.ap.beta1, self.ap.beta2, self.ap.eps = beta1, beta2, eps
        self.ap.grad_scale = 1.0

    def set_lr(self, lr_planes):
        lr = np.asarray(lr_planes, np.float32).reshape(-1)
        if lr.shape[0] != NPLANES:
            raise ValueError("need one learning rate per parameter plane (59)")
        self.ap.lr[:] = [float(x) for x in lr]

    def begin_step(self, grad_scale: float = 1.0):
        """Open the next iteration for `apply_planes` (several partial updates that share one step count)."""
        self.step_count += 1
        self.ap.step = self.step_count
        self.ap.grad_scale = float(grad_scale)

    def apply_planes(self, grads: torch.Tensor, plane0: int, n_planes: int):
        m = self.model
        L.check(L.load().omfs_adam_step_planes(L.ptr(m.params), L.ptr(grads), L.ptr(self.m), L.ptr(self.v), m.n, m.n_pad,
                                               C.byref(self.ap), int(plane0), int(n_planes), L.stream_ptr()), "omfs_adam_step_planes")

    def apply_sh_rest(self, drgb: torch.Tensor, view_dir: torch.Tensor, sh_degree: int, grads_low: torch.Tensor | None = None):
        """The step on the 45 SH planes of degree >= 1 with their gradient Y_k(dir) * drgb formed where it is consumed
        (omfs_adam_step_sh_rest; drgb, view_dir [3][n_pad] as omfs_project_bwd left them): those gradient planes are never written.
        grads_low: the gradient buffer -- planes 0..13 are then updated from it in the same launch."""
        m = self.model
        L.check(L.load().omfs_adam_step_sh_rest(L.ptr(m.params), L.ptr(grads_low), L.ptr(drgb), L.ptr(view_dir), L.ptr(self.m), L.ptr(self.v),
                                                m.n, m.n_pad, C.byref(self.ap), int(sh_degree), L.stream_ptr()), "omfs_adam_step_sh_rest")

    def apply_range(self, grad_shard: torch.Tensor, offset: int, count: int):
        """The update on the flat range [offset, offset + count) of the [59][n_pad] buffers; grad_shard holds that range's
        (summed) gradient.  Data-parallel "sharded" exchange: moments outside the rank's range are not touched."""
        mdl = self.model
        p, m, v = mdl.params.view(-1), self.m.view(-1), self.v.view(-1)
        L.check(L.load().omfs_adam_step_range(L.ptr(p[offset:]), L.ptr(grad_shard), L.ptr(m[offset:]), L.ptr(v[offset:]), mdl.n_pad,
                                              int(offset), int(count), C.byref(self.ap), L.stream_ptr()), "omfs_adam_step_range")

    def step(self, grads: torch.Tensor, grad_scale: float = 1.0):
        self.step_count += 1
        self.ap.step = self.step_count
        self.ap.grad_scale = float(grad_scale)
        m = self.model
        L.check(L.load().omfs_adam_step(L.ptr(m.params), L.ptr(grads), L.ptr(self.m), L.ptr(self.v), m.n, m.n_pad,
                                        C.byref(self.ap), L.stream_ptr()), "omfs_adam_step")


def default_lr_planes(position_lr=1.6e-4, scaling_lr=1.7e-2, rotation_lr=1e-3, opacity_lr=5e-2, feature_lr=2.5e-3):
    """Per-plane learning rates (GaussianAvatars' larger scaling lr)."""
    lr = np.zeros(NPLANES, np.float32)
    lr[0:3] = position_lr
    lr[3:6] = scaling_lr
    lr[6:10] = rotation_lr
    lr[10] = opacity_lr
    lr[11:14] = feature_lr
    lr[14:] = feature_lr / 20.0
    return lr
