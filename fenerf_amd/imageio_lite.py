"""Image / video writing for the inference front ends without torchvision, cv2 or skvideo (none of them is a dependency of
this package): the subset of `torchvision.utils.make_grid` / `save_image` the reference's scripts use
(render_multiview_images_double_semantic.py:84-85, render_video_interpolation_semantic.py:290-312, :411-458), PNG through
Pillow, and an uncompressed AVI writer for `--save_with_video`.

make_grid semantics kept (torchvision 0.9, the reference's pin): images laid out row-major, `nrow` images per row, `padding`
pixels of `pad_value` around every image; `normalize=True` maps [low, high] (`value_range`, or the min / max of the WHOLE
batch) to [0, 1] after clamping; single-channel images are repeated to three channels."""
import math
import struct

import numpy as np
import torch


def _grid_np(tensor, nrow=8, padding=2, normalize=False, value_range=None, pad_value=0.0):
    """make_grid on the host in numpy float32 (the same fp32 operations in the same order as the torch statements it replaced: clamp,
    - low, / (high - low); single-threaded: a 128 x 128 image takes 0.1 ms wherever it runs, where the chain of small torch CPU operations
    took 4.7 ms per image in a container granted 16 of its 256 logical CPUs -- tools/exp/dump_timing.py)."""
    t = torch.as_tensor(tensor).detach()
    t = np.asarray((t if t.dtype == torch.float32 else t.float()).cpu().numpy(), dtype=np.float32)
    if t.ndim == 2:
        t = t[None]
    if t.ndim == 3:
        t = t[None]
    if t.shape[1] == 1:
        t = np.repeat(t, 3, axis=1)
    if normalize:
        low, high = (float(value_range[0]), float(value_range[1])) if value_range is not None else (float(t.min()), float(t.max()))
        t = (np.clip(t, np.float32(low), np.float32(high)) - np.float32(low)) / np.float32(max(high - low, 1e-5))
    B, C, H, W = t.shape
    if B == 1:
        return t[0]
    xmaps = min(nrow, B)
    ymaps = int(math.ceil(B / xmaps))
    hh, ww = H + padding, W + padding
    grid = np.full((C, hh * ymaps + padding, ww * xmaps + padding), np.float32(pad_value), dtype=np.float32)
    for k in range(B):
        y, x = divmod(k, xmaps)
        grid[:, y * hh + padding: y * hh + padding + H, x * ww + padding: x * ww + padding + W] = t[k]
    return grid


def make_grid(tensor, nrow=8, padding=2, normalize=False, value_range=None, pad_value=0.0):
    """[B,C,H,W] (or [C,H,W] / [H,W]) float tensor -> [3,Hg,Wg] grid (a single image is returned unpadded, like torchvision)."""
    return torch.from_numpy(np.ascontiguousarray(_grid_np(tensor, nrow, padding, normalize, value_range, pad_value)))


def to_uint8_hwc(grid):
    """[3,H,W] in [0,1] -> uint8 [H,W,3], rounded like torchvision.utils.save_image (x*255 + 0.5, clamp, truncate)."""
    g = grid.detach().cpu().numpy() if isinstance(grid, torch.Tensor) else grid
    g = np.clip(np.asarray(g, dtype=np.float32) * np.float32(255) + np.float32(0.5), np.float32(0), np.float32(255))
    return np.ascontiguousarray(g.transpose(1, 2, 0).astype(np.uint8))


def save_image(tensor, path, nrow=8, padding=2, normalize=False, value_range=None, pad_value=0.0):
    """torchvision.utils.save_image for PNG files (Pillow)."""
    from PIL import Image
    arr = to_uint8_hwc(_grid_np(tensor, nrow=nrow, padding=padding, normalize=normalize, value_range=value_range, pad_value=pad_value))
    Image.fromarray(arr).save(path)
    return arr


JET = None


def jet_colormap(gray_u8):
    """uint8 [H,W] -> uint8 RGB [H,W,3]: the piecewise-linear 'jet' ramp (what cv2.COLORMAP_JET approximates; the reference
    colours depth maps with it, render_video_interpolation_semantic.py:424-427).  Not bit-identical to OpenCV's 256-entry table."""
    global JET
    if JET is None:
        x = np.arange(256) / 255.0
        r = np.clip(1.5 - np.abs(4 * x - 3), 0, 1)
        g = np.clip(1.5 - np.abs(4 * x - 2), 0, 1)
        b = np.clip(1.5 - np.abs(4 * x - 1), 0, 1)
        JET = (np.stack([r, g, b], -1) * 255 + 0.5).astype(np.uint8)
    return JET[np.asarray(gray_u8, dtype=np.uint8)]


class AviWriter:
    """Uncompressed RGB AVI (RIFF 'AVI ' with one 'vids' stream of BI_RGB 24-bit DIB frames, bottom-up BGR) -- playable by
    ffmpeg / VLC / OpenCV, written with the standard library only.  Frames are uint8 [H,W,3] RGB of constant size."""

    def __init__(self, path, fps=25):
        self.path, self.fps, self.frames, self.size = path, int(fps), [], None

    def write(self, rgb):
        rgb = np.ascontiguousarray(np.asarray(rgb, dtype=np.uint8))
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        if self.size is None:
            self.size = rgb.shape[:2]
        assert rgb.shape[:2] == self.size, "all frames must have the same size"
        H, W = self.size
        row = (W * 3 + 3) // 4 * 4
        buf = np.zeros((H, row), np.uint8)
        buf[:, :W * 3] = rgb[::-1, :, ::-1].reshape(H, W * 3)        # bottom-up, BGR
        self.frames.append(buf.tobytes())

    def release(self):
        if not self.frames:
            return
        H, W = self.size
        n, fsz = len(self.frames), len(self.frames[0])
        chunk = lambda tag, data: tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
        lst = lambda tag, data: b"LIST" + struct.pack("<I", len(data) + 4) + tag + data
        avih = struct.pack("<14I", 1000000 // self.fps, fsz * self.fps, 0, 0x10, n, 0, 1, fsz, W, H, 0, 0, 0, 0)
        strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"DIB ", 0, 0, 0, 0, 1, self.fps, 0, n, fsz, 0xFFFFFFFF, 0, 0, 0, W, H)
        strf = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, fsz, 0, 0, 0, 0)
        hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
        movi_data = b"".join(chunk(b"00db", f) for f in self.frames)
        idx, off = b"", 4
        for f in self.frames:
            idx += struct.pack("<4sIII", b"00db", 0x10, off, len(f))
            off += 8 + len(f) + (len(f) & 1)
        body = b"AVI " + hdrl + lst(b"movi", movi_data) + chunk(b"idx1", idx)
        with open(self.path, "wb") as fh:
            fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        self.frames = []


def write_mrc(path, volume):
    """A float32 volume [nz, ny, nx] as an MRC2014 map (mode 2) -- what the reference's shape scripts write through
    `mrcfile.new_mmap(path, shape=volume.shape, mrc_mode=2); mrc.data[:] = volume` (extract_double_semantic_shapes.py:120-121; mrcfile is
    not a dependency here).  1024-byte little-endian header in mrcfile's defaults for a new volume: nx / ny / nz = mx / my / mz = the
    array's axes reversed, cell = one unit per voxel, 90-degree angles, axis order (1, 2, 3), ISPG 1, density statistics of the
    data, 'MAP ' + machine stamp 0x44 0x44, NVERSION 20140.  Readable by mrcfile, ChimeraX and read_mrc below."""
    vol = np.ascontiguousarray(volume, dtype="<f4")
    if vol.ndim != 3:
        raise ValueError(f"write_mrc: a 3-d volume, got shape {vol.shape}")
    nz, ny, nx = vol.shape
    h = bytearray(1024)
    struct.pack_into("<3i", h, 0, nx, ny, nz)
    struct.pack_into("<i", h, 12, 2)                                   # MODE 2: 32-bit float
    struct.pack_into("<3i", h, 16, 0, 0, 0)                            # nxstart ..
    struct.pack_into("<3i", h, 28, nx, ny, nz)                         # mx, my, mz
    struct.pack_into("<3f", h, 40, float(nx), float(ny), float(nz))    # cella: voxel size 1
    struct.pack_into("<3f", h, 52, 90.0, 90.0, 90.0)                   # cellb
    struct.pack_into("<3i", h, 64, 1, 2, 3)                            # mapc, mapr, maps
    dmin, dmax, dmean = (float(vol.min()), float(vol.max()), float(vol.mean(dtype=np.float64))) if vol.size else (0.0, 0.0, 0.0)
    rms = float(np.sqrt(((vol.astype(np.float64) - dmean) ** 2).mean())) if vol.size else 0.0
    struct.pack_into("<3f", h, 76, dmin, dmax, dmean)
    struct.pack_into("<i", h, 88, 1)                                   # ISPG 1: a volume (0 would be an image stack)
    struct.pack_into("<i", h, 92, 0)                                   # NSYMBT
    h[104:108] = b"MRCO"                                               # EXTTYP (mrcfile's default)
    struct.pack_into("<i", h, 108, 20140)                              # NVERSION
    h[208:212] = b"MAP "
    h[212:216] = bytes([0x44, 0x44, 0x00, 0x00])                       # machine stamp: little-endian
    struct.pack_into("<f", h, 216, rms)
    label = b"fenerf_amd.imageio_lite.write_mrc"
    struct.pack_into("<i", h, 220, 1)                                  # NLABL
    h[224:224 + len(label)] = label
    with open(path, "wb") as f:
        f.write(bytes(h))
        f.write(vol.tobytes())


def read_mrc(path):
    """(volume [nz, ny, nx] float32, header dict) of a mode-2 MRC file (the inverse of write_mrc; tests and the marching-cubes step of a
    downstream tool)."""
    raw = open(path, "rb").read()
    nx, ny, nz, mode = struct.unpack_from("<4i", raw, 0)
    if raw[208:212] != b"MAP " or mode != 2:
        raise ValueError(f"{path}: not a mode-2 MRC map")
    nsymbt = struct.unpack_from("<i", raw, 92)[0]
    data = np.frombuffer(raw, "<f4", nx * ny * nz, 1024 + nsymbt).reshape(nz, ny, nx)
    dmin, dmax, dmean = struct.unpack_from("<3f", raw, 76)
    return data, dict(nx=nx, ny=ny, nz=nz, mode=mode, cella=struct.unpack_from("<3f", raw, 40), ispg=struct.unpack_from("<i", raw, 88)[0],
                      dmin=dmin, dmax=dmax, dmean=dmean, rms=struct.unpack_from("<f", raw, 216)[0], nversion=struct.unpack_from("<i", raw, 108)[0])
