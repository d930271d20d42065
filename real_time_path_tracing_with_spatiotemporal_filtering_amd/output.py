"""Output path standing in for the swapchain blit (main.cpp:1338-1361): the reference's
only output is the RGBA32F `image` blitted to an 8-bit swapchain image (VK_FORMAT_B8G8R8A8_UNORM: the blit
clamps to [0,1] and quantises, no tone curve).  `to_unorm8` is that conversion; `tonemap` adds the optional
exposure/gamma a viewer wants; `write_png` / `write_pfm` need nothing beyond numpy + zlib."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def to_unorm8(image: np.ndarray) -> np.ndarray:
    """float RGB(A) -> uint8 RGB exactly like a UNORM blit: clamp to [0,1], round(x*255); NaN -> 0."""
    rgb = np.nan_to_num(np.asarray(image, np.float32)[..., :3], nan=0.0, posinf=1.0, neginf=0.0)
    return (np.clip(rgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)


def tonemap(image: np.ndarray, exposure: float = 1.0, gamma: float = 2.2) -> np.ndarray:
    """viewer transform (not in the reference): exposure, clamp, gamma-encode -> uint8 RGB"""
    rgb = np.nan_to_num(np.asarray(image, np.float32)[..., :3], nan=0.0, posinf=1e30, neginf=0.0) * np.float32(exposure)
    return to_unorm8(np.clip(rgb, 0.0, 1.0) ** np.float32(1.0 / gamma))


def write_png(path: str, rgb8: np.ndarray) -> None:
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w, c = rgb8.shape
    if c != 3:
        raise ValueError("write_png expects HxWx3 uint8")
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb8.reshape(h, w * 3)], axis=1).tobytes()  # filter 0 per row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def write_pfm(path: str, image: np.ndarray) -> None:
    """little-endian colour PFM, bottom row first (the format the C++ host's --dump writes)"""
    rgb = np.ascontiguousarray(np.asarray(image, np.float32)[::-1, :, :3])
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.astype("<f4").tobytes())


def read_pfm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError("not a colour PFM")
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return np.ascontiguousarray(data[::-1]).astype(np.float32)
