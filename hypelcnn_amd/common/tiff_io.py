"""Minimal baseline-TIFF writer for the result rasters (the reference uses tifffile.imwrite,
classify/infer_for_classification.py:67-68; tifffile is not available in this image).

Uncompressed, little-endian, one strip: uint8 grayscale [H,W] or RGB [H,W,3]."""
import struct

import numpy


def imwrite(path, image):
    img = numpy.ascontiguousarray(image)
    if img.dtype != numpy.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
        raise ValueError("imwrite: uint8 [H,W] or [H,W,3] expected")
    h, w = img.shape[:2]
    spp = 1 if img.ndim == 2 else 3
    data = img.tobytes()
    entries = []

    def tag(code, typ, count, value):
        entries.append(struct.pack("<HHII", code, typ, count, value))

    n_tags = 10 if spp == 1 else 10
    ifd_off = 8 + len(data) + (len(data) & 1)
    extra_off = ifd_off + 2 + n_tags * 12 + 4
    tag(256, 4, 1, w)                      # ImageWidth
    tag(257, 4, 1, h)                      # ImageLength
    if spp == 1:
        tag(258, 3, 1, 8)                  # BitsPerSample
    else:
        tag(258, 3, 3, extra_off)          # -> three shorts after the IFD
    tag(259, 3, 1, 1)                      # no compression
    tag(262, 3, 1, 1 if spp == 1 else 2)   # BlackIsZero / RGB
    tag(273, 4, 1, 8)                      # StripOffsets
    tag(277, 3, 1, spp)                    # SamplesPerPixel
    tag(278, 4, 1, h)                      # RowsPerStrip
    tag(279, 4, 1, len(data))              # StripByteCounts
    tag(284, 3, 1, 1)                      # PlanarConfiguration: chunky
    with open(path, "wb") as f:
        f.write(b"II*\x00" + struct.pack("<I", ifd_off))
        f.write(data)
        if len(data) & 1:
            f.write(b"\x00")
        f.write(struct.pack("<H", len(entries)) + b"".join(entries) + struct.pack("<I", 0))
        if spp == 3:
            f.write(struct.pack("<HHH", 8, 8, 8))


def imread(path):
    """Reads back what imwrite wrote (and any other uncompressed, single-strip, 8-bit chunky little-endian TIFF)."""
    raw = open(path, "rb").read()
    if raw[:4] != b"II*\x00":
        raise ValueError("imread: little-endian baseline TIFF expected")
    ifd = struct.unpack_from("<I", raw, 4)[0]
    n = struct.unpack_from("<H", raw, ifd)[0]
    tags = {}
    for i in range(n):
        code, typ, count, value = struct.unpack_from("<HHII", raw, ifd + 2 + 12 * i)
        tags[code] = (typ, count, value)
    if tags.get(259, (0, 0, 1))[2] != 1:
        raise ValueError("imread: compressed TIFF not supported")
    w, h, spp = tags[256][2], tags[257][2], tags.get(277, (0, 0, 1))[2]
    off, cnt = tags[273][2], tags[279][2]
    img = numpy.frombuffer(raw, numpy.uint8, cnt, off)
    return img.reshape(h, w) if spp == 1 else img.reshape(h, w, spp)
