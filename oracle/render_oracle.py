"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/render_oracle.c (CPU oracle of the synthesis path)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "librender_oracle.so")


class Scene(ctypes.Structure):
    _fields_ = [("hand_verts", ctypes.c_void_p), ("hand_faces", ctypes.c_void_p), ("hand_normals", ctypes.c_void_p),
                ("hand_uv", ctypes.c_void_p), ("hand_map", ctypes.c_void_p), ("hand_tex", ctypes.c_void_p), ("hts", ctypes.c_int),
                ("obj_verts", ctypes.c_void_p), ("obj_normals", ctypes.c_void_p), ("obj_uv", ctypes.c_void_p),
                ("obj_faces", ctypes.c_void_p), ("obj_vert_off", ctypes.c_void_p), ("obj_face_off", ctypes.c_void_p),
                ("obj_tex", ctypes.c_void_p), ("ots", ctypes.c_int), ("bg", ctypes.c_void_p), ("bgs", ctypes.c_int),
                ("srgb2lin", ctypes.c_void_p), ("lin2srgb", ctypes.c_void_p),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
                ("W", ctypes.c_int), ("H", ctypes.c_int)]


SAMPLE_DTYPE = np.dtype([("obj_id", "<i4"), ("hand_tex_id", "<i4"), ("bg_id", "<i4"), ("bg_x0", "<i4"), ("bg_y0", "<i4"),
                         ("bg_w", "<i4"), ("bg_h", "<i4"), ("light", "<f4"), ("obj_pose", "<f4", (16,))])

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "render_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(SO)
    return _lib


def color_luts():
    """sRGB <-> linear tables shared (as data) by the oracle and the HIP renderer."""
    c = np.arange(256, dtype=np.float64) / 255.0
    s2l = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)
    l = np.arange(4096, dtype=np.float64) / 4095.0
    l2s = np.where(l <= 0.0031308, l * 12.92, 1.055 * l ** (1 / 2.4) - 0.055)
    return s2l, np.clip(np.floor(l2s * 255.0 + 0.5), 0, 255).astype(np.uint8)


class SceneHolder:
    """Keeps the numpy arrays alive and exposes the C struct."""

    def __init__(self, assets, fx=435.0, fy=435.0, cx=256.0, cy=256.0, W=512, H=512):
        h = assets.hand
        self.arrs = dict(
            hand_faces=np.ascontiguousarray(h["faces"], np.int32), hand_normals=np.ascontiguousarray(h["normals"], np.float32),
            hand_uv=np.ascontiguousarray(h["uv"], np.float32), hand_map=np.ascontiguousarray(h["map"], np.int32),
            hand_tex=np.ascontiguousarray(assets.hand_tex),
            obj_verts=np.ascontiguousarray(assets.obj_verts, np.float32), obj_normals=np.ascontiguousarray(assets.obj_normals, np.float32),
            obj_uv=np.ascontiguousarray(assets.obj_uv, np.float32), obj_faces=np.ascontiguousarray(assets.obj_faces, np.int32),
            obj_vert_off=np.ascontiguousarray(assets.obj_vert_off, np.int32), obj_face_off=np.ascontiguousarray(assets.obj_face_off, np.int32),
            obj_tex=np.ascontiguousarray(assets.obj_tex), bg=np.ascontiguousarray(assets.backgrounds))
        s2l, l2s = color_luts()
        self.arrs["srgb2lin"], self.arrs["lin2srgb"] = s2l, l2s
        sc = Scene()
        for k, v in self.arrs.items():
            setattr(sc, k, v.ctypes.data)
        sc.hts, sc.ots, sc.bgs = assets.hand_tex.shape[1], assets.obj_tex.shape[1], assets.backgrounds.shape[1]
        sc.fx, sc.fy, sc.cx, sc.cy, sc.W, sc.H = fx, fy, cx, cy, W, H
        self.sc = sc
        self.W, self.H = W, H

    def rasterize(self, sample, hand_verts):
        keys = np.empty(self.W * self.H, np.uint64)
        hv = np.ascontiguousarray(hand_verts, np.float32)
        lib().ro_rasterize(ctypes.byref(self.sc), sample.ctypes.data_as(ctypes.c_void_p), hv.ctypes.data_as(ctypes.c_void_p),
                           keys.ctypes.data_as(ctypes.c_void_p))
        return keys.reshape(self.H, self.W)

    def shade(self, sample, hand_verts, keys):
        img = np.empty((self.H, self.W, 4), np.uint8)
        hv = np.ascontiguousarray(hand_verts, np.float32)
        lib().ro_shade(ctypes.byref(self.sc), sample.ctypes.data_as(ctypes.c_void_p), hv.ctypes.data_as(ctypes.c_void_p),
                       np.ascontiguousarray(keys).ctypes.data_as(ctypes.c_void_p), img.ctypes.data_as(ctypes.c_void_p))
        return img

    def render_batch(self, samples, hand_verts, order, factor, inv_affine, ow, oh, blur=None):
        B = len(samples)
        out = np.empty((B, 3, oh, ow), np.float32)
        rgbx = np.empty((B, self.H, self.W, 4), np.uint8)
        keys = np.empty((B, self.H, self.W), np.uint64)
        hv = np.ascontiguousarray(hand_verts, np.float32)
        order = np.ascontiguousarray(order, np.int32)
        factor = np.ascontiguousarray(factor, np.float32)
        inv = np.ascontiguousarray(inv_affine, np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        blur = None if blur is None else np.ascontiguousarray(blur, np.float32)
        lib().ro_render_batch(ctypes.byref(self.sc), p(samples), p(hv), ctypes.c_int(B), p(order), p(factor), p(inv),
                              None if blur is None else p(blur), ctypes.c_int(ow), ctypes.c_int(oh), p(out), p(rgbx), p(keys))
        return out, rgbx, keys


def color_jitter(rgbx, order, factor):
    img = np.ascontiguousarray(rgbx, np.uint8).copy()
    o = np.ascontiguousarray(order, np.int32)
    f = np.ascontiguousarray(factor, np.float32)
    lib().ro_color_jitter(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(img.shape[0] * img.shape[1]),
                          o.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p))
    return img


def affine_crop(rgbx, inv, ow, oh):
    img = np.ascontiguousarray(rgbx, np.uint8)
    out = np.empty((3, oh, ow), np.float32)
    inv = np.ascontiguousarray(inv, np.float32)
    lib().ro_affine_crop(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(img.shape[1]), ctypes.c_int(img.shape[0]),
                         inv.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(ow), ctypes.c_int(oh),
                         out.ctypes.data_as(ctypes.c_void_p))
    return out


def gaussian_blur(rgbx, radius):
    """PIL ImageFilter.GaussianBlur(radius) on the RGB bytes of an RGBX image (H, W, 4)."""
    img = np.ascontiguousarray(rgbx, np.uint8).copy()
    lib().ro_gaussian_blur(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(img.shape[1]), ctypes.c_int(img.shape[0]),
                           ctypes.c_float(radius))
    return img


def augment(rgb, order, factor, inv, blur, flip, ow, oh):
    """HOdata.__getitem__'s image chain (hodata.py:336-337,435-446) on one uint8 (H, W, 3) frame -> float32 [3, oh, ow]."""
    rgb = np.asarray(rgb, np.uint8)
    img = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    img[..., :3] = rgb
    o = np.ascontiguousarray(order, np.int32); f = np.ascontiguousarray(factor, np.float32)
    inv = np.ascontiguousarray(inv, np.float32)
    out = np.empty((3, oh, ow), np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    lib().ro_augment(p(img), ctypes.c_int(img.shape[1]), ctypes.c_int(img.shape[0]), ctypes.c_int(int(bool(flip))),
                     ctypes.c_float(blur), p(o), p(f), p(inv), ctypes.c_int(ow), ctypes.c_int(oh), p(out))
    return out
