#!/usr/bin/env python3
"""Minimal read-only parser of a TensorFlow-Lite flatbuffer (no tflite / flatbuffers / tensorflow package needed; nothing
from the file is executed): tensors (shape, type, quantisation, constant data) and the operator list of subgraph 0.

Used by tools/gen_policy_fixture.py to turn the policy the reference ships
(/root/reference/src/balance_robot/envs/RobotMovePolicy.tflite, loaded at ref:envs/RobotMoveBaseEnv.py:81-98) into a plain
weights fixture.  Field numbers follow the public TFLite schema (schema.fbs, v3):
  Model{version, operator_codes, subgraphs, description, buffers}; SubGraph{tensors, inputs, outputs, operators, name};
  Tensor{shape, type, buffer, name, quantization}; QuantizationParameters{min, max, scale, zero_point, details_type, details,
  quantized_dimension}; Operator{opcode_index, inputs, outputs, builtin_options_type, builtin_options};
  OperatorCode{deprecated_builtin_code, custom_code, version, builtin_code}; Buffer{data}.
"""
import struct
import numpy as np

TENSOR_TYPES = {0: np.float32, 1: np.float16, 2: np.int32, 3: np.uint8, 4: np.int64, 6: np.bool_, 7: np.int16, 9: np.int8, 10: np.float64}
# the builtin operators a small MLP export can contain (BuiltinOperator enum of the public schema)
BUILTIN = {0: "ADD", 2: "CONCATENATION", 6: "DEQUANTIZE", 9: "FULLY_CONNECTED", 14: "LOGISTIC", 18: "MUL", 19: "RELU", 22: "RESHAPE",
           25: "SOFTMAX", 28: "TANH", 34: "PAD", 39: "TRANSPOSE", 40: "MEAN", 41: "SUB", 42: "DIV", 47: "EXP", 49: "SPLIT", 53: "CAST",
           55: "MAXIMUM", 57: "MINIMUM", 59: "NEG", 73: "LOG", 74: "SUM", 75: "SQRT", 76: "RSQRT", 77: "SHAPE", 78: "POW", 83: "PACK",
           88: "UNPACK", 92: "SQUARE", 94: "FILL", 97: "RESIZE_NEAREST_NEIGHBOR", 99: "ABS", 114: "QUANTIZE", 117: "HARD_SWISH",
           45: "STRIDED_SLICE", 36: "GATHER", 65: "SLICE", 70: "EXPAND_DIMS", 43: "SQUEEZE", 3: "CONV_2D", 84: "LOGICAL_OR",
           61: "GREATER", 58: "LESS", 64: "SELECT", 123: "SELECT_V2", 101: "RANGE", 102: "RESIZE_BILINEAR"}


class _Table:
    def __init__(self, buf, pos):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        self.vt_len = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, field):
        o = 4 + 2 * field
        if o >= self.vt_len:
            return 0
        return struct.unpack_from("<H", self.buf, self.vt + o)[0]

    def scalar(self, field, fmt, default=0):
        o = self._off(field)
        return struct.unpack_from("<" + fmt, self.buf, self.pos + o)[0] if o else default

    def _indirect(self, field):
        o = self._off(field)
        if not o:
            return None
        p = self.pos + o
        return p + struct.unpack_from("<I", self.buf, p)[0]

    def table(self, field):
        p = self._indirect(field)
        return _Table(self.buf, p) if p is not None else None

    def string(self, field):
        p = self._indirect(field)
        if p is None:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n]).decode("utf-8", "replace")

    def vector(self, field, fmt):
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        return list(struct.unpack_from(f"<{n}{fmt}", self.buf, p + 4)) if n else []

    def bytes_(self, field):
        p = self._indirect(field)
        if p is None:
            return b""
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n])

    def tables(self, field):
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for i in range(n):
            q = p + 4 + 4 * i
            out.append(_Table(self.buf, q + struct.unpack_from("<I", self.buf, q)[0]))
        return out


def read_tflite(path):
    """-> dict(version, description, tensors=[dict(name, shape, dtype, scale, zero_point, qdim, data or None)], inputs, outputs,
    operators=[dict(op, inputs, outputs, fused_activation)])"""
    buf = memoryview(open(path, "rb").read())
    if bytes(buf[4:8]) != b"TFL3":
        raise ValueError("not a TFLite flatbuffer (identifier TFL3 missing)")
    model = _Table(buf, struct.unpack_from("<I", buf, 0)[0])
    codes = []
    for oc in model.tables(1):
        dep, code = oc.scalar(0, "b"), oc.scalar(3, "i")
        codes.append(max(dep, code))
    buffers = [b.bytes_(0) for b in model.tables(4)]
    sg = model.tables(2)[0]
    tensors = []
    for t in sg.tables(0):
        ttype = t.scalar(1, "b")
        dt = TENSOR_TYPES.get(ttype)
        shape = t.vector(0, "i")
        raw = buffers[t.scalar(2, "I")]
        q = t.table(4)
        scale = q.vector(2, "f") if q is not None else []
        zp = q.vector(3, "q") if q is not None else []
        qdim = q.scalar(6, "i") if q is not None else 0
        data = None
        if raw and dt is not None:
            data = np.frombuffer(raw, dtype=dt).reshape(shape if shape else ()).copy()
        tensors.append(dict(name=t.string(3), shape=shape, dtype=np.dtype(dt).name if dt is not None else f"type{ttype}", scale=scale,
                            zero_point=zp, qdim=qdim, data=data))
    ops = []
    for o in sg.tables(3):
        code = codes[o.scalar(0, "I")]
        name = BUILTIN.get(code, f"OP{code}")
        fused = None
        if name == "FULLY_CONNECTED":
            bo = o.table(4)
            fused = {0: None, 1: "RELU", 2: "RELU_N1_TO_1", 3: "RELU6", 4: "TANH"}.get(bo.scalar(0, "b") if bo is not None else 0)
        ops.append(dict(op=name, inputs=o.vector(1, "i"), outputs=o.vector(2, "i"), fused_activation=fused))
    return dict(version=model.scalar(0, "I"), description=model.string(3), tensors=tensors, inputs=sg.vector(1, "i"),
                outputs=sg.vector(2, "i"), operators=ops)


if __name__ == "__main__":
    import sys
    m = read_tflite(sys.argv[1])
    print("version", m["version"], "|", m["description"], "| inputs", m["inputs"], "outputs", m["outputs"])
    for i, t in enumerate(m["tensors"]):
        print(f"  t{i:3d} {t['dtype']:8s} {str(t['shape']):12s} scale {t['scale'][:2]}{'...' if len(t['scale']) > 2 else ''} zp {t['zero_point'][:2]} "
              f"{'const' if t['data'] is not None else '     '} {t['name']}")
    for o in m["operators"]:
        print("  ", o["op"], o["inputs"], "->", o["outputs"], o["fused_activation"] or "")
