"""Test-only WRITERS of the checkpoint container formats (ONNX protobuf wire format, Kaldi matrices): the product package only
reads these formats (vadx.onnx_reader, vadx.checkpoints); the tests need files to read and no package that writes them exists in
this image.  tests/test_checkpoints.py additionally decodes byte strings assembled BY HAND from the format specifications, so that
a mistake made symmetrically in a reader and in these writers cannot hide."""
import struct

import numpy as np

_DTYPE_CODES = {np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3, np.dtype(np.int32): 6, np.dtype(np.int64): 7,
                np.dtype(np.float16): 10, np.dtype(np.float64): 11}


def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(field, value):
    return _enc_varint((field << 3) | 0) + _enc_varint(value)


def enc_tensor(name, arr, raw=True):
    a = np.ascontiguousarray(arr)
    code = _DTYPE_CODES[a.dtype]
    body = b"".join(_vi(1, int(d)) for d in a.shape) + _vi(2, code) + _ld(8, name.encode())
    if raw or a.dtype not in (np.float32, np.int64):
        body += _ld(9, a.astype(a.dtype.newbyteorder("<")).tobytes())
    elif a.dtype == np.float32:
        body += _ld(4, a.astype("<f4").tobytes())                     # packed float_data
    else:
        body += _ld(7, b"".join(_enc_varint(int(v)) for v in a.reshape(-1)))
    return body


def enc_node(op_type, inputs, outputs, name="", attrs=None):
    """attrs: name -> int | float | bytes | ndarray (tensor) | ('graph', bytes) | list of ints"""
    body = b"".join(_ld(1, s.encode()) for s in inputs) + b"".join(_ld(2, s.encode()) for s in outputs)
    if name:
        body += _ld(3, name.encode())
    body += _ld(4, op_type.encode())
    for k, v in (attrs or {}).items():
        a = _ld(1, k.encode())
        if isinstance(v, tuple) and v[0] == "graph":
            a += _ld(6, v[1]) + _vi(20, 5)
        elif isinstance(v, np.ndarray):
            a += _ld(5, enc_tensor("", v)) + _vi(20, 4)
        elif isinstance(v, float):
            a += _enc_varint((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)
        elif isinstance(v, (bytes, str)):
            a += _ld(4, v if isinstance(v, bytes) else v.encode()) + _vi(20, 3)
        elif isinstance(v, (list, tuple)):
            a += b"".join(_vi(8, int(x)) for x in v) + _vi(20, 7)
        else:
            a += _vi(3, int(v)) + _vi(20, 2)
        body += _ld(5, a)
    return body


def enc_graph(nodes=(), initializers=(), name="g"):
    """nodes: encoded NodeProto bodies; initializers: (name, array[, raw]) tuples -> encoded GraphProto body"""
    body = b"".join(_ld(1, n) for n in nodes) + _ld(2, name.encode())
    for init in initializers:
        body += _ld(5, enc_tensor(*init))
    return body


def write_onnx(path, graph_body, ir_version=8, opset=16):
    model = _vi(1, ir_version) + _ld(7, graph_body) + _ld(8, _ld(1, b"") + _vi(2, opset))
    with open(path, "wb") as fh:
        fh.write(model)
    return path


def write_kaldi_matrix(path, mat, key="global", binary=True):
    """Inverse of read_kaldi_matrix (used by the tests and for exporting statistics)."""
    m = np.ascontiguousarray(mat)
    with open(path, "wb") as fh:
        if binary:
            tok = b"DM " if m.dtype == np.float64 else b"FM "
            if tok == b"FM ":
                m = m.astype(np.float32)
            fh.write(key.encode() + b" \0B" + tok + b"\4" + struct.pack("<i", m.shape[0]) + b"\4" + struct.pack("<i", m.shape[1]))
            fh.write(m.tobytes())
        else:
            body = "\n".join("  " + " ".join(repr(float(v)) for v in row) for row in m)
            fh.write((key + "  [\n" + body + " ]\n").encode())


