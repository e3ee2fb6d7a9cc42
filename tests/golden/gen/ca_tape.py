"""Decoder/evaluator for the CasADi-3.6 serialized SXFunction files (*.ca) that the
reference ships under bound_planner/RobotModel/ (fk_pos, fk_pos_col_0..5, hom_trans,
jacobian).  Used ONLY by the golden-vector generator in this container; it reads
/root/reference and therefore never runs on the GPU box.

File format (probed):
  text  -> bytes : byte = (ord(c0)-97) | (ord(c1)-97) << 4
  tape           : n_instr records of 16 bytes directly before a trailer of
                   26 + 9*nnz_out bytes; record = int32 op, int32 i0, int32 i1, int32 i2
                   except OP_CONST (44): int32 op, int32 dst, double value.
  ops            : 1 ADD 2 SUB 3 MUL 4 DIV 5 NEG 11 SQ 12 TWICE 13 SIN 14 COS
                   44 CONST 45 INPUT(dst,arg,nz) 46 OUTPUT(arg,src,nz)
Outputs are dense column-major.
"""
import struct

import numpy as np

_VALID = {1, 2, 3, 4, 5, 11, 12, 13, 14, 44, 45, 46}


def _bytes(path):
    s = open(path).read().strip()
    return bytes(((ord(s[i]) - 97) | ((ord(s[i + 1]) - 97) << 4)) for i in range(0, len(s) - 1, 2))


def load_tape(path, out_shape):
    """Return list of (op, a, b, c|value) instructions."""
    raw = _bytes(path)
    nnz = int(np.prod(out_shape))
    end = len(raw) - (26 + 9 * nnz)
    instrs = []
    pos = end - 16
    n_out = 0
    while pos >= 0:
        op, i0 = struct.unpack_from("<ii", raw, pos)
        if op not in _VALID:
            break
        if op == 44:
            (val,) = struct.unpack_from("<d", raw, pos + 8)
            instrs.append((op, i0, val, 0))
        else:
            i1, i2 = struct.unpack_from("<ii", raw, pos + 8)
            instrs.append((op, i0, i1, i2))
        if op == 46:
            n_out += 1
        pos -= 16
    instrs.reverse()

    # The backward scan can swallow header bytes that happen to look like valid records;
    # the true tape is the longest suffix in which every operand is defined before use.
    def consistent(seq):
        defined = set()
        for op, a, b, c in seq:
            if op == 44 or op == 45:
                defined.add(a)
            elif op == 46:
                if b not in defined:
                    return False
            else:
                if b not in defined or (op in (1, 2, 3, 4) and c not in defined):
                    return False
                defined.add(a)
        return True

    start = 0
    while not consistent(instrs[start:]):
        start += 1
    instrs = instrs[start:]
    assert sum(1 for ins in instrs if ins[0] == 46) == nnz, (path, n_out, nnz)
    return instrs


class TapeFunction:
    def __init__(self, path, out_shape):
        self.out_shape = tuple(out_shape)
        self.instrs = load_tape(path, out_shape)
        self.n_work = 1 + max(
            max(ins[1] for ins in self.instrs if ins[0] != 46),
            max(ins[2] for ins in self.instrs if ins[0] == 46),
        )

    def __call__(self, q):
        q = np.asarray(q).reshape(-1)
        dt = complex if np.iscomplexobj(q) else float
        w = np.zeros(self.n_work, dtype=dt)
        out = np.zeros(int(np.prod(self.out_shape)), dtype=dt)
        for op, a, b, c in self.instrs:
            if op == 44:
                w[a] = b
            elif op == 45:
                w[a] = q[c]
            elif op == 46:
                out[c] = w[b]
            elif op == 1:
                w[a] = w[b] + w[c]
            elif op == 2:
                w[a] = w[b] - w[c]
            elif op == 3:
                w[a] = w[b] * w[c]
            elif op == 4:
                w[a] = w[b] / w[c]
            elif op == 5:
                w[a] = -w[b]
            elif op == 11:
                w[a] = w[b] * w[b]
            elif op == 12:
                w[a] = 2 * w[b]
            elif op == 13:
                w[a] = np.sin(w[b])
            elif op == 14:
                w[a] = np.cos(w[b])
            else:
                raise ValueError(op)
        return out.reshape(self.out_shape, order="F")


REF_MODEL_DIR = "/root/reference/bound_planner/RobotModel/"


def load_all(model_dir=REF_MODEL_DIR):
    f = {
        "fk_pos": TapeFunction(model_dir + "fk_pos.ca", (3, 1)),
        "hom_trans": TapeFunction(model_dir + "hom_trans.ca", (4, 4)),
        "jacobian": TapeFunction(model_dir + "jacobian.ca", (6, 7)),
    }
    for i in range(6):
        f[f"fk_pos_col_{i}"] = TapeFunction(model_dir + f"fk_pos_col_{i}.ca", (3, 1))
    return f


if __name__ == "__main__":
    fs = load_all()
    q = np.array([0, 0, 0, -np.pi / 2, 0, np.pi / 2, 0.0])
    for k, f in fs.items():
        print(k, len(f.instrs), f.n_work)
    print(fs["fk_pos"](q).ravel())
    print(fs["hom_trans"](q))
