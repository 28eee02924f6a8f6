"""Minimal ONNX container reader: the weights (initialisers and Constant-node tensors) and the node list of a `.onnx` file,
straight from the protobuf wire format -- neither `onnx` nor `onnxruntime` is needed (or installed here).

Why it exists: the Silero network never appears in the reference tree -- the reference copies the pip
package's `silero_vad/data/silero_vad.onnx` and feeds it to onnxruntime (Silero/Export_Silero_VAD.py:91,
Silero/modeling_modified/utils_vad.py:117).  Reading that file's initialisers is the only way to run (and to pin) the
real weights on the HIP path.  Only the message fields a weight extractor needs are decoded:

    ModelProto.graph(7) -> GraphProto{ node(1), initializer(5) }
    NodeProto{ input(1) output(2) name(3) op_type(4) attribute(5) }
    AttributeProto{ name(1) f(2) i(3) s(4) t(5) g(6) ints(8) graphs(11) }        (g / graphs: If / Loop bodies, walked recursively)
    TensorProto{ dims(1) data_type(2) float_data(4) int32_data(5) int64_data(7) name(8) raw_data(9) double_data(10) }

`write_onnx` is the inverse for the same subset (used by the tests to build synthetic containers, and handy for exporting
a vadx weight dict back into a file onnxruntime could open).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_,
           10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# --------------------------------------------------------------------------------------------- wire format, reading
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf):
    """yield (field number, wire type, value) over one message; value = int (varint / fixed) or memoryview (length-delimited)"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            if len(v) != ln:
                raise ValueError("truncated length-delimited field")
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield field, wt, v


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


def _tensor(buf):
    dims, dtype, name, raw = [], 1, "", None
    floats, int32s, int64s, doubles = [], [], [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            dims += _packed_varints(v) if wt == 2 else [_signed(v)]
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode("utf-8")
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.frombuffer(v, "<f4"))
        elif f == 10:
            doubles.append(np.frombuffer(bytes(v), "<f8") if wt == 2 else np.frombuffer(v, "<f8"))
        elif f == 5:
            int32s += _packed_varints(v) if wt == 2 else [_signed(v)]
        elif f == 7:
            int64s += _packed_varints(v) if wt == 2 else [_signed(v)]
        elif f == 14 and v == 1:
            raise ValueError(f"tensor {name!r}: external data files are not supported")
    if dtype not in _DTYPES:
        raise ValueError(f"tensor {name!r}: unsupported ONNX data type {dtype}")
    dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dtype=dt.newbyteorder("<"))
    elif floats:
        arr = np.concatenate(floats).astype(dt)
    elif doubles:
        arr = np.concatenate(doubles).astype(dt)
    elif int64s:
        arr = np.array(int64s, dtype=np.int64).astype(dt)
    elif int32s:
        a = np.array(int32s, dtype=np.int64)
        arr = a.astype(np.uint16).view(np.float16) if dt == np.float16 else a.astype(dt)      # fp16 rides as its bit pattern
    else:
        arr = np.zeros(0, dtype=dt)
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        raise ValueError(f"tensor {name!r}: {arr.size} values for dims {dims}")
    return name, np.array(arr, dtype=dt).reshape(dims)


class Node:
    __slots__ = ("scope", "op_type", "name", "inputs", "outputs", "attrs")

    def __init__(self, scope):
        self.scope, self.op_type, self.name, self.inputs, self.outputs, self.attrs = scope, "", "", [], [], {}

    def __repr__(self):
        return f"Node({self.scope!r}, {self.op_type}, {self.name!r}, in={self.inputs}, out={self.outputs})"


class OnnxGraph:
    """tensors: OrderedDict (scope, name) -> ndarray; scope = () for the main graph, (("If_0", "then_branch"),) for a sub-graph,
    one (node tag, attribute name) pair per nesting level; Constant-node tensors are keyed by the node's output name.
    nodes: every NodeProto of every (sub)graph, `node.scope` likewise."""

    def __init__(self):
        self.tensors = OrderedDict()
        self.nodes = []

    def scopes(self):
        seen = []
        for sc, _ in self.tensors:
            if sc not in seen:
                seen.append(sc)
        for n in self.nodes:
            if n.scope not in seen:
                seen.append(n.scope)
        return seen

    def in_scope(self, scope):
        """name -> tensor visible from `scope`: its own and those of every enclosing scope (inner names shadow outer ones)"""
        out = OrderedDict()
        for depth in range(len(scope) + 1):
            for (sc, name), v in self.tensors.items():
                if sc == scope[:depth]:
                    out[name] = v
        return out

    def nodes_in(self, scope):
        return [n for n in self.nodes if n.scope == scope]


def _graph(buf, scope, g):
    counter = 0
    for f, _wt, v in _fields(buf):
        if f == 5:
            name, arr = _tensor(v)
            g.tensors[(scope, name)] = arr
        elif f == 1:
            node = Node(scope)
            attrs_raw = []
            for nf, _nwt, nv in _fields(v):
                if nf == 1:
                    node.inputs.append(bytes(nv).decode("utf-8"))
                elif nf == 2:
                    node.outputs.append(bytes(nv).decode("utf-8"))
                elif nf == 3:
                    node.name = bytes(nv).decode("utf-8")
                elif nf == 4:
                    node.op_type = bytes(nv).decode("utf-8")
                elif nf == 5:
                    attrs_raw.append(nv)
            tag = node.name or f"{node.op_type}_{counter}"
            counter += 1
            for a in attrs_raw:
                an, val = "", None
                sub = []
                for af, awt, av in _fields(a):
                    if af == 1:
                        an = bytes(av).decode("utf-8")
                    elif af == 2:
                        val = struct.unpack("<f", av)[0]
                    elif af == 3:
                        val = _signed(av)
                    elif af == 4:
                        val = bytes(av)
                    elif af == 5:
                        val = _tensor(av)[1]
                    elif af == 8:
                        val = (val or []) + (_packed_varints(av) if awt == 2 else [_signed(av)])
                    elif af in (6, 11):
                        sub.append(av)
                for k, sg in enumerate(sub):
                    _graph(sg, scope + ((tag, an + (f"[{k}]" if len(sub) > 1 else "")),), g)
                if val is not None:
                    node.attrs[an] = val
            if node.op_type == "Constant" and isinstance(node.attrs.get("value"), np.ndarray) and node.outputs:
                g.tensors[(scope, node.outputs[0])] = node.attrs["value"]
            g.nodes.append(node)
    return g


def read_onnx(path_or_bytes):
    """-> OnnxGraph of a `.onnx` file (path or bytes)."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = memoryview(bytes(path_or_bytes))
    else:
        with open(path_or_bytes, "rb") as fh:
            data = memoryview(fh.read())
    g = OnnxGraph()
    found = False
    for f, wt, v in _fields(data):
        if f == 7 and wt == 2:
            _graph(v, (), g)
            found = True
    if not found:
        raise ValueError("not an ONNX ModelProto: no graph field")
    return g
