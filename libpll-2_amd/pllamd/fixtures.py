"""(De)serialise a driver.Case plus expected results as one .npz golden fixture (tests/golden/).
A fixture is data only: inputs of the hot path and the outputs the reference produced for them."""
import json

import numpy as np

from .driver import Case

_ARRAYS = ("pmatrix", "freqs", "charmap", "tip_clvs", "rate_weights", "pattern_weights", "prop_invar",
           "freqs_indices", "asc_weights")
_SCALARS = ("name", "states", "rate_cats", "tips", "sites", "attributes", "clv_buffers", "scale_buffers",
            "update_repeats", "asc_type")


def save(path, case: Case, expected: dict, extra: dict = None, arrays: dict = None):
    """arrays: additional named numpy arrays (e.g. an eigensystem, expected derivatives); they come
    back from load() as extra['arrays']"""
    meta = {k: getattr(case, k) for k in _SCALARS}
    meta["op_batches"] = [[list(map(int, op)) for op in b] for b in case.op_batches]
    meta["edges"] = [list(map(int, e)) for e in case.edges]
    meta["roots"] = [list(map(int, e)) for e in case.roots]
    meta["dump_clvs"] = None if case.dump_clvs is None else list(map(int, case.dump_clvs))
    meta["extra"] = extra or {}
    arrs = {}
    for k in _ARRAYS:
        v = getattr(case, k)
        if v is not None:
            arrs["in_" + k] = np.asarray(v)
    # branches that share a length share a matrix: store each distinct matrix once
    pm = arrs.pop("in_pmatrix")
    uniq, inverse = np.unique(pm.reshape(pm.shape[0], -1), axis=0, return_inverse=True)
    arrs["in_pmatrix_unique"] = uniq.reshape((-1,) + pm.shape[1:])
    arrs["in_pmatrix_index"] = inverse.astype(np.uint32).reshape(-1)
    if case.sequences is not None:
        arrs["in_sequences"] = np.stack([np.frombuffer(s, dtype=np.uint8) for s in case.sequences])
    for idx, a in expected["clv"].items():
        arrs[f"out_clv_{idx}"] = a
    for idx, a in expected["scaler"].items():
        arrs[f"out_scaler_{idx}"] = a
    arrs["out_lnl"] = np.asarray(expected["lnl"], dtype=np.float64)
    if expected["persite"]:
        arrs["out_persite"] = np.stack(expected["persite"])
    arrs["out_root_lnl"] = np.asarray(expected.get("root_lnl", []), dtype=np.float64)
    if expected.get("root_persite"):
        arrs["out_root_persite"] = np.stack(expected["root_persite"])
    for k, v in (arrays or {}).items():
        arrs["x_" + k] = np.asarray(v)
    arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)


def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    kw = {k: meta[k] for k in _SCALARS if k in meta}
    for k in _ARRAYS:
        if "in_" + k in z:
            kw[k] = z["in_" + k]
    kw["pmatrix"] = z["in_pmatrix_unique"][z["in_pmatrix_index"]]
    if "in_sequences" in z:
        kw["sequences"] = [row.tobytes() for row in z["in_sequences"]]
    kw["op_batches"] = [[tuple(op) for op in b] for b in meta["op_batches"]]
    kw["edges"] = [tuple(e) for e in meta["edges"]]
    kw["roots"] = [tuple(e) for e in meta["roots"]]
    kw["dump_clvs"] = meta["dump_clvs"]
    case = Case(**kw)
    exp = {"clv": {}, "scaler": {}, "lnl": list(z["out_lnl"]), "persite": [], "root_lnl": list(z["out_root_lnl"]),
           "root_persite": []}
    for k in z.files:
        if k.startswith("out_clv_"):
            exp["clv"][int(k[8:])] = z[k]
        elif k.startswith("out_scaler_"):
            exp["scaler"][int(k[11:])] = z[k]
    if "out_persite" in z:
        exp["persite"] = list(z["out_persite"])
    if "out_root_persite" in z:
        exp["root_persite"] = list(z["out_root_persite"])
    extra = dict(meta["extra"])
    extra["arrays"] = {k[2:]: z[k] for k in z.files if k.startswith("x_")}
    return case, exp, extra
