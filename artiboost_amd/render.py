"""Device-side scene + batched renderer (HIP) -- the data plane of the reference's Renderer / RendererProvider /
RenderedDataset image path (anakin/utils/renderer.py:44-136, anakin/artiboost/render_infra.py,
anakin/artiboost/rendered_dataset.py:256-270)."""
import ctypes

import numpy as np
import torch

from . import _lib as L

SAMPLE_DTYPE = np.dtype([("obj_id", "<i4"), ("hand_tex_id", "<i4"), ("bg_id", "<i4"), ("bg_x0", "<i4"), ("bg_y0", "<i4"),
                         ("bg_w", "<i4"), ("bg_h", "<i4"), ("light", "<f4"), ("obj_pose", "<f4", (16,))])


class _Scene(ctypes.Structure):
    _fields_ = [("hand_faces", ctypes.c_void_p), ("hand_normals", ctypes.c_void_p), ("hand_uv", ctypes.c_void_p),
                ("hand_map", ctypes.c_void_p), ("hand_tex", ctypes.c_void_p), ("hts", ctypes.c_int), ("obj_verts", ctypes.c_void_p),
                ("obj_normals", ctypes.c_void_p), ("obj_uv", ctypes.c_void_p), ("obj_faces", ctypes.c_void_p),
                ("obj_vert_off", ctypes.c_void_p), ("obj_face_off", ctypes.c_void_p), ("obj_tex", ctypes.c_void_p),
                ("ots", ctypes.c_int), ("bg", ctypes.c_void_p), ("bgs", ctypes.c_int), ("srgb2lin", ctypes.c_void_p),
                ("lin2srgb", ctypes.c_void_p), ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float),
                ("cy", ctypes.c_float), ("W", ctypes.c_int), ("H", ctypes.c_int)]


def color_luts():
    c = np.arange(256, dtype=np.float64) / 255.0
    s2l = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)
    l = np.arange(4096, dtype=np.float64) / 4095.0
    l2s = np.where(l <= 0.0031308, l * 12.92, 1.055 * l ** (1 / 2.4) - 0.055)
    return s2l, np.clip(np.floor(l2s * 255.0 + 0.5), 0, 255).astype(np.uint8)


def _rgbx(rgb):
    """uint8 [..., 3] -> [..., 4]: the background texels are fetched as aligned dwords."""
    a = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    a[..., :3] = rgb
    return a


class DeviceRenderer:
    """Renderer(width, height).setup(cam_intr, obj_meshes, hand_meshes, backgrounds) on the GPU; `render()` draws a
    whole batch (the reference's __call__ draws one image per Python call)."""

    def __init__(self, assets, cam_intr, width=512, height=512, device="cuda"):
        self.dev = torch.device(device)
        self.W, self.H = width, height
        h = assets.hand
        s2l, l2s = color_luts()
        host = dict(hand_faces=np.ascontiguousarray(h["faces"], np.int32), hand_normals=h["normals"], hand_uv=h["uv"],
                    hand_map=np.ascontiguousarray(h["map"], np.int32),
                    hand_tex=assets.hand_tex, obj_verts=assets.obj_verts, obj_normals=assets.obj_normals,
                    obj_uv=assets.obj_uv, obj_faces=assets.obj_faces, obj_vert_off=assets.obj_vert_off,
                    obj_face_off=assets.obj_face_off, obj_tex=assets.obj_tex, bg=_rgbx(assets.backgrounds), srgb2lin=s2l,
                    lin2srgb=l2s)
        self.t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.dev) for k, v in host.items()}
        sc = _Scene()
        for k, v in self.t.items():
            setattr(sc, k, v.data_ptr())
        sc.hts, sc.ots, sc.bgs = assets.hand_tex.shape[1], assets.obj_tex.shape[1], assets.backgrounds.shape[1]
        sc.fx, sc.fy, sc.cx, sc.cy = [float(x) for x in (cam_intr[0, 0], cam_intr[1, 1], cam_intr[0, 2], cam_intr[1, 2])]
        sc.W, sc.H = width, height
        self.sc = sc
        self.max_faces = 1538 + assets.max_obj_faces
        self._ws = None

    def render(self, samples_dev, hand_verts, order, factor, inv_affine, ow, oh, out_pad=None, out_chw=None,
               want_keys=False, want_rgbx=False, blur=None, pad_code=None):
        """samples_dev: uint8 device tensor [B,96] (SAMPLE_DTYPE records); hand_verts [B,778,3] f32; order int32 [B,4];
        factor f32 [B,4]; inv_affine f32 [B,6]; blur: f32 [B] GaussianBlur radii (< 1.41) or None.
        Returns dict(keys=..., rgbx=...) of the optional outputs (rgbx: the render before blur and jitter)."""
        B = hand_verts.shape[0]
        lib = L.lib()
        need = lib.ab_render_workspace_bytes(L.i(B), L.i(self.W), L.i(self.H), L.i(self.max_faces))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        keys = torch.empty((B, self.H, self.W), dtype=torch.int64, device=self.dev) if want_keys else None
        rgbx = torch.empty((B, self.H, self.W, 4), dtype=torch.uint8, device=self.dev) if want_rgbx else None
        dt = L.dt(out_pad) if out_pad is not None else 0
        if pad_code is not None:          # 2 = AB_DT_U8N: out_pad (bf16) receives the integer plane 2 v - 255
            if pad_code == 2 and (out_pad is None or out_pad.dtype != torch.bfloat16):
                raise TypeError("AB_DT_U8N writes a bfloat16 padded image")
            dt = pad_code
        L.check(lib.ab_render_batch(ctypes.byref(self.sc), L.ptr(samples_dev), L.ptr(hand_verts), L.ptr(order),
                                    L.ptr(factor), L.ptr(inv_affine), L.ptr(blur), L.i(B), L.i(self.max_faces), L.i(ow), L.i(oh),
                                    L.i(dt), L.ptr(out_pad), L.ptr(out_chw), L.ptr(self._ws), L.ptr(keys), L.ptr(rgbx),
                                    L.stream()), "ab_render_batch")
        return dict(keys=keys, rgbx=rgbx)
