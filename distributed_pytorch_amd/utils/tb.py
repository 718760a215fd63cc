"""Minimal TensorBoard scalar writer (tensorboard is not installed on the targets).

Produces real ``events.out.tfevents.*`` files — TFRecord framing with masked
CRC32-C and hand-encoded ``Event``/``Summary`` protobufs — plus a ``scalars.jsonl``
mirror.  API subset of ``torch.utils.tensorboard.SummaryWriter`` that the
reference uses (``imagenet_ddp_apex.py:154-159,279-293``): ``SummaryWriter(comment=)``,
``.log_dir``, ``.add_scalar(tag, value, step)``, ``.close()``; the run directory is
named ``runs/<time>_<host><comment>`` like upstream.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import time
from datetime import datetime

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag: str, value: float, step: int, wall_time: float) -> bytes:
    val = _field_bytes(1, tag.encode("utf-8")) + _varint((2 << 3) | 5) + struct.pack("<f", value)
    summary = _field_bytes(1, val)
    return (_varint((1 << 3) | 1) + struct.pack("<d", wall_time) +
            _varint((2 << 3) | 0) + _varint(step & 0xFFFFFFFFFFFFFFFF) + _field_bytes(5, summary))


def encode_version_event(wall_time: float) -> bytes:
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _field_bytes(3, b"brain.Event:2")


def tfrecord(payload: bytes) -> bytes:
    header = struct.pack("<Q", len(payload))
    return (header + struct.pack("<I", _masked_crc(header)) + payload +
            struct.pack("<I", _masked_crc(payload)))


class SummaryWriter:
    def __init__(self, log_dir: str = None, comment: str = ""):
        if not log_dir:
            stamp = datetime.now().strftime("%b%d_%H-%M-%S")
            log_dir = os.path.join("runs", stamp + "_" + socket.gethostname() + comment)
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        name = "events.out.tfevents.%010d.%s.%d.0" % (int(time.time()), socket.gethostname(),
                                                     os.getpid())
        self._f = open(os.path.join(log_dir, name), "wb")
        self._f.write(tfrecord(encode_version_event(time.time())))
        self._jsonl = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag: str, scalar_value, global_step: int = 0, walltime: float = None):
        v = float(scalar_value)
        t = time.time() if walltime is None else walltime
        self._f.write(tfrecord(encode_scalar_event(tag, v, int(global_step), t)))
        self._jsonl.write(json.dumps({"tag": tag, "value": v, "step": int(global_step),
                                      "wall_time": t}) + "\n")

    def flush(self):
        self._f.flush()
        self._jsonl.flush()

    def close(self):
        self.flush()
        self._f.close()
        self._jsonl.close()
