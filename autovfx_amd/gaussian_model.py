"""Gaussian parameter store with the reference's getters and PLY format.

Host-side mirror of what the render path reads from ``GaussianModel``
(``sugar/gaussian_splatting/scene/gaussian_model.py``): raw parameters ``_xyz, _features_dc [P,1,3],
_features_rest [P,M-1,3], _scaling, _rotation, _opacity`` and the activated getters (``:95-128``:
``exp``, ``normalize``, ``sigmoid``, ``cat(dc, rest)``, ``get_normal`` via ``get_minimum_axis`` /
``flip_align_view``, ``utils/general_utils.py:78-101,135-157``).  Training-side machinery (optimizer,
densification, ``simple_knn`` initialisation) is out of scope.

PLY I/O follows ``save_ply`` / ``load_ply`` (``gaussian_model.py:187-266``) without the ``plyfile``
dependency (absent here): binary little-endian, one ``vertex`` element, float32 properties
``x y z nx ny nz f_dc_0..2 f_rest_0..(3(M-1)-1) opacity scale_0..2 rot_0..3``, with ``f_dc`` / ``f_rest`` stored
channel-major (``[P,3,K]`` flattened), i.e. transposed relative to the in-memory ``[P,K,3]``.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """Quaternion (w,x,y,z, any norm) -> rotation matrix [P,3,3] (``general_utils.py:78-101``)."""
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device, dtype=r.dtype)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def build_covariance_from_scaling_rotation(scaling: torch.Tensor, scaling_modifier: float, rotation: torch.Tensor) -> torch.Tensor:
    """The six upper-triangle entries of R S S^T R^T, [P,6] (``scene/gaussian_model.py:26-30`` with ``build_scaling_rotation`` /
    ``strip_symmetric``, ``general_utils.py:64-76,101-110``): what ``render()`` hands over as ``cov3D_precomp`` when
    ``pipe.compute_cov3D_python`` is set (``gaussian_renderer/__init__.py:127-128``).  Note that the reference passes the RAW rotation
    (``self._rotation``, ``gaussian_model.py:114``): ``build_rotation`` normalises it itself."""
    s = scaling_modifier * scaling
    L = torch.zeros((s.shape[0], 3, 3), dtype=torch.float, device=s.device)
    R = build_rotation(rotation)
    L[:, 0, 0], L[:, 1, 1], L[:, 2, 2] = s[:, 0], s[:, 1], s[:, 2]
    L = R @ L
    cov = L @ L.transpose(1, 2)
    out = torch.zeros((cov.shape[0], 6), dtype=torch.float, device=s.device)
    out[:, 0], out[:, 1], out[:, 2] = cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2]
    out[:, 3], out[:, 4], out[:, 5] = cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]
    return out


def get_minimum_axis(scales: torch.Tensor, rotations: torch.Tensor) -> torch.Tensor:
    """The rotation column belonging to the smallest scale (``general_utils.py:135-141``; the reference
    flags its own implementation as questionable but it is what ships, so it is mirrored as is)."""
    sorted_idx = torch.argsort(scales, descending=False, dim=-1)
    R = build_rotation(rotations)
    R_sorted = torch.gather(R, dim=2, index=sorted_idx[:, None, :].repeat(1, 3, 1)).squeeze()
    return R_sorted[:, :, 0]


def flip_align_view(normal: torch.Tensor, viewdir: torch.Tensor):
    """Flip normals to face the viewer (``general_utils.py:151-157``)."""
    dotprod = torch.sum(normal * -viewdir, dim=-1, keepdims=True)
    non_flip = dotprod >= 0
    return normal * torch.where(non_flip, 1, -1), non_flip


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.log(x / (1 - x))


class GaussianModel:
    """Parameters + getters; ``active_sh_degree`` / ``max_sh_degree`` as in the reference."""

    def __init__(self, sh_degree: int = 3):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e

    # ---- construction ----
    @staticmethod
    def from_activated(means3D, opacities, scales, rotations, shs, sh_degree: int = 3) -> "GaussianModel":
        """Build from activated tensors (a ``scenes.GaussianCloud``): inverts the activations."""
        m = GaussianModel(sh_degree)
        m._xyz = means3D.clone()
        m._features_dc = shs[:, :1].clone().contiguous()
        m._features_rest = shs[:, 1:].clone().contiguous()
        m._scaling = torch.log(scales)
        m._rotation = rotations.clone()
        m._opacity = inverse_sigmoid(opacities.clamp(1e-6, 1 - 1e-6))
        m.active_sh_degree = sh_degree
        return m

    def to(self, device) -> "GaussianModel":
        for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            setattr(self, k, getattr(self, k).to(device))
        self.__dict__.pop("_memo_cache", None)
        return self

    # ---- getters (gaussian_model.py:95-128) ----
    # The reference recomputes every activation on every frame (at C3 the SH concat alone copies 1.1 GB).
    # The results only depend on the parameters, so while autograd is off they are memoised on the
    # parameters' identity + in-place version; any optimizer step or edit invalidates them.  The entry keeps the
    # parameter tensors it was computed from alive, so `is` cannot be fooled by a recycled id / address.
    # Contract: a write that does not go through the tensor's version counter (`param.data.add_()`, a raw-pointer
    # write from another extension) is invisible here -- call `invalidate_cache()` after such a write.
    def invalidate_cache(self) -> None:
        self.__dict__.pop("_memo_cache", None)

    memoise = True   # False: every getter recomputes on every call, as the reference's getters do (bench.py times that shape)

    def _memo(self, name, deps, fn):
        if not self.memoise or (torch.is_grad_enabled() and any(t.requires_grad for t in deps)):
            return fn()
        versions = tuple((t._version, t.data_ptr(), tuple(t.shape)) for t in deps)
        cache = self.__dict__.setdefault("_memo_cache", {})
        hit = cache.get(name)
        if hit is None or len(hit[0]) != len(deps) or any(a is not b for a, b in zip(hit[0], deps)) or hit[1] != versions:
            with torch.no_grad():
                value = fn()
            ready = None
            if value.is_cuda:   # consumers on other HIP streams (multi-stream rendering) must wait for the producer
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(value.device))
            hit = (tuple(deps), versions, value, ready)
            cache[name] = hit
        elif hit[3] is not None:
            cur = torch.cuda.current_stream(hit[2].device)
            cur.wait_event(hit[3])
            hit[2].record_stream(cur)   # the allocator must not recycle the block while this stream still reads it
        return hit[2]

    @property
    def get_scaling(self):
        return self._memo("scaling", (self._scaling,), lambda: torch.exp(self._scaling))

    @property
    def get_rotation(self):
        return self._memo("rotation", (self._rotation,), lambda: torch.nn.functional.normalize(self._rotation))

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return self._memo("features", (self._features_dc, self._features_rest),
                          lambda: torch.cat((self._features_dc, self._features_rest), dim=1))

    @property
    def get_opacity(self):
        return self._memo("opacity", (self._opacity,), lambda: torch.sigmoid(self._opacity))

    @property
    def get_minimum_axis(self):
        # (made contiguous once here: the strided column view would be copied by every frame's kernel call)
        return self._memo("min_axis", (self._scaling, self._rotation),
                          lambda: get_minimum_axis(self.get_scaling, self.get_rotation).contiguous())

    def get_normal(self, dir_pp_normalized=None):
        normal_axis, _ = flip_align_view(self.get_minimum_axis, dir_pp_normalized)
        return normal_axis / normal_axis.norm(dim=1, keepdim=True)

    def get_covariance(self, scaling_modifier: float = 1):
        """``GaussianModel.get_covariance`` (``scene/gaussian_model.py:113-114``)."""
        return build_covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    # ---- PLY (gaussian_model.py:187-266) ----
    def _attribute_names(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["opacity"] + [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path: str) -> None:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        f32 = lambda t: t.detach().cpu().numpy().astype(np.float32)
        xyz = f32(self._xyz)
        cols = [xyz, np.zeros_like(xyz),
                f32(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
                f32(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
                f32(self._opacity), f32(self._scaling), f32(self._rotation)]
        table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
        names = self._attribute_names()
        assert table.shape[1] == len(names)
        header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
        header += [f"property float {n}" for n in names] + ["end_header"]
        with open(path, "wb") as f:
            f.write(("\n".join(header) + "\n").encode("ascii"))
            f.write(table.tobytes())

    def load_ply(self, path: str, device: Optional[str] = None) -> "GaussianModel":
        names, table = read_ply_vertex_table(path)
        col = {n: i for i, n in enumerate(names)}
        pick = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
        rest_names = pick("f_rest_")
        assert len(rest_names) == 3 * (self.max_sh_degree + 1) ** 2 - 3, "PLY SH count does not match max_sh_degree"
        # the whole table goes to the device in ONE copy; the columns are picked (and f_rest transposed) there
        tbl = torch.from_numpy(np.ascontiguousarray(table)).to(device=device, dtype=torch.float32)

        def cols(keys):
            idx = [col[k] for k in keys]
            if idx == list(range(idx[0], idx[0] + len(idx))):          # the usual layout: a contiguous block of columns
                return tbl[:, idx[0]:idx[0] + len(idx)]
            return tbl[:, torch.as_tensor(idx, device=tbl.device)]

        self._xyz = cols(["x", "y", "z"]).contiguous()
        self._opacity = cols(["opacity"]).contiguous()
        self._scaling = cols(pick("scale_")).contiguous()
        self._rotation = cols(pick("rot")).contiguous()
        self._features_dc = cols(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(-1, 3, 1).transpose(1, 2).contiguous()
        self._features_rest = cols(rest_names).reshape(-1, 3, (self.max_sh_degree + 1) ** 2 - 1).transpose(1, 2).contiguous()
        self.active_sh_degree = self.max_sh_degree
        self.invalidate_cache()
        return self

    # the six entries of a SuGaR checkpoint's ``state_dict`` that ``scene_representation.load_scene`` reads (:200-205)
    _SUGAR_KEYS = {"_xyz": "_points", "_opacity": "all_densities", "_features_dc": "_sh_coordinates_dc",
                   "_features_rest": "_sh_coordinates_rest", "_scaling": "_scales", "_rotation": "_quaternions"}

    def load_sugar_checkpoint(self, path: str, device: Optional[str] = None) -> "GaussianModel":
        """Parameters of a (coarse or refined) SuGaR ``.pt`` checkpoint, the way ``scene_representation.py:192-214``
        takes them: six tensors out of ``checkpoint['state_dict']``, stored raw (pre-activation) exactly like a PLY's
        columns, ``active_sh_degree = max_sh_degree``."""
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        state = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
        missing = [k for k in self._SUGAR_KEYS.values() if k not in state]
        if missing:
            raise KeyError(f"{path}: not a SuGaR checkpoint, state_dict lacks {missing}")
        for attr, key in self._SUGAR_KEYS.items():
            setattr(self, attr, state[key].detach().to(dtype=torch.float32, device=device).contiguous())
        P = self._xyz.shape[0]
        if self._features_dc.shape != (P, 1, 3) or self._features_rest.dim() != 3 or self._features_rest.shape[::2] != (P, 3):
            raise ValueError(f"{path}: SH tensors have shapes {tuple(self._features_dc.shape)}, "
                             f"{tuple(self._features_rest.shape)}; expected [P,1,3] and [P,K,3]")
        # No check of K against max_sh_degree, as in the reference: its SuGaR hparams say max_sh_degree = 4 for
        # checkpoints holding 16 coefficients (sh_levels = 4), and the rasterizer evaluates degrees above 3 as 3.
        self.active_sh_degree = self.max_sh_degree
        self.__dict__.pop("_memo_cache", None)
        return self


def load_scene(path: str, max_sh_degree: int = 4, device: Optional[str] = None) -> GaussianModel:
    """``scene_representation.load_scene`` (:192-221): a SuGaR ``.pt`` checkpoint is read with ``max_sh_degree``
    (SuGaR's hparams use 4: the reference's own comment "SuGaR: 4, vanilla 3DGS: 3"), a vanilla 3DGS ``.ply`` with
    ``max_sh_degree - 1``."""
    if path.endswith(".pt"):
        return GaussianModel(max_sh_degree).load_sugar_checkpoint(path, device)
    if path.endswith(".ply"):
        return GaussianModel(max_sh_degree - 1).load_ply(path, device)
    raise ValueError(f"{path}: expected a SuGaR .pt checkpoint or a 3DGS .ply")


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_vertex_table(path: str):
    """Minimal PLY reader: the ``vertex`` element of a binary-little-endian or ASCII file as a table (float32 when every
    property is a 4-byte float, float64 otherwise) plus its property names (what ``plyfile`` gives ``load_ply``)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif props:
                    pass  # elements after the vertex block are ignored
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [n for n, _ in props]
        if fmt == "binary_little_endian":
            rec = np.dtype([(n, t) for n, t in props])
            raw = bytearray(rec.itemsize * count)      # (writable: torch.from_numpy wants to own what it wraps)
            got = f.readinto(raw)
            if got != len(raw):
                raise ValueError(f"{path}: the vertex element is truncated ({got} of {len(raw)} bytes)")
            if count and all(np.dtype(t) == np.dtype("<f4") for _n, t in props):
                # every property a 4-byte float (what 3DGS writes): the records ARE a float32 table -- no per-column conversion
                # (62 strided column copies to float64 made a 1 M-Gaussian scene take 0.9 s to read, a 60 k object 40 - 120 ms)
                table = np.frombuffer(raw, dtype="<f4", count=count * len(props)).reshape(count, len(props))
            else:
                data = np.frombuffer(raw, dtype=rec, count=count)
                table = np.stack([data[n].astype(np.float64) for n in names], axis=1) if count else np.zeros((0, len(names)))
        elif fmt == "ascii":
            table = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2) if count else np.zeros((0, len(names)))
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return names, table
