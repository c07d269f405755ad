"""Mirror of the `pixsfm._pixsfm._features` containers the adjusters touch
(pixsfm/features/bindings.cc:38-300): FeaturePatch -> FeatureMap (per image) -> FeatureSet
(per level) -> FeatureManager, plus Reference.  They are thin host-side holders of numpy
views (the reference's patches built from numpy are non-owning views too,
features/src/featurepatch.cc:45); the hot path consumes the flat HBM arena built from them
(`to_arena`).  HDF5 loading / caching is out of scope.
"""
from collections.abc import Mapping as _Mapping

import warnings
import weakref

import numpy as np

kDenseId = 1000000      # util/src/types.h:33


class _PatchStatus:
    """features/src/featurepatch.h PatchStatus (bindings.cc:260-262)."""

    def __init__(self, is_locked=True, reference_count=1):
        self.is_locked, self.reference_count = bool(is_locked), int(reference_count)


class _CallableList(list):
    def __call__(self):
        return self


class TrackElementTuple(tuple):
    """colmap::TrackElement as the reference exposes it (.image_id, .point2D_idx) that also IS the (image_id, point2D_idx)
    tuple this package used before."""

    def __new__(cls, image_id, point2D_idx):
        return super().__new__(cls, (int(image_id), int(point2D_idx)))

    def __getnewargs__(self):                    # pickle / copy re-create through __new__(cls, image_id, point2D_idx)
        return (self[0], self[1])

    image_id = property(lambda self: self[0])
    point2D_idx = property(lambda self: self[1])


class TrackList(list):
    """colmap::Track surface (.elements, .length()) over a list of track elements."""

    elements = property(lambda self: list(self))

    def length(self):
        return len(self)


class FeaturePatch:
    """features/src/featurepatch.h:40-156: HWC data + corner (x0, y0) + scale (sx, sy)."""

    def __init__(self, inarray=None, offset=(0, 0), scale=(1.0, 1.0), do_copy=False, data=None, corner=None):
        """(inarray, offset, scale, do_copy) like the pybind constructor (features/bindings.cc:47-53: `offset` is the corner
        (x0, y0)); `data` / `corner` are this package's own names for the first two."""
        data = inarray if data is None else data
        corner = offset if corner is None else corner
        data = np.array(data) if do_copy else np.asarray(data)
        if data.ndim != 3:
            raise ValueError("FeaturePatch expects an H x W x C array")
        if data.dtype not in (np.float16, np.float32, np.float64):
            raise ValueError("FeaturePatch dtype must be float16/32/64")     # featurepatch.cc:365-367
        self.data = np.ascontiguousarray(data)
        self._ptr = self.data.ctypes.data            # cached: to_arena reads it once per patch
        self.corner = np.asarray(corner, dtype=np.int32).reshape(2)
        self.scale = np.asarray(scale, dtype=np.float64).reshape(2)
        self.corner.setflags(write=False); self.scale.setflags(write=False)     # read-only like the pybind properties (features/bindings.cc:55-58)
        self._meta = (self._ptr, int(self.corner[0]), int(self.corner[1]), float(self.scale[0]), float(self.scale[1]),
                      self.data.shape + (self.data.dtype.str,))

    @property
    def shape(self):
        return self.data.shape

    # -- the read-only properties / helpers of the pybind class (features/bindings.cc:47-76) --
    height = property(lambda self: int(self.data.shape[0]))
    width = property(lambda self: int(self.data.shape[1]))
    channels = property(lambda self: int(self.data.shape[2]))
    size = property(lambda self: int(self.data.size))

    @property
    def status(self):
        """PatchStatus of a patch built from numpy: locked, one reference (featurepatch.cc:41-43); nothing is ever unloaded."""
        return _PatchStatus(True, 1)

    @property
    def upsampling_factor(self):                 # featurepatch.h: 1 unless a cost-map extractor set it
        return getattr(self, "_upsampling_factor", 1.0)

    @upsampling_factor.setter
    def upsampling_factor(self, value):
        self._upsampling_factor = float(value)

    def get_entry(self, y, x, c):                # featurepatch.cc GetEntry: data[(y W + x) C + c]
        return self.data[int(y), int(x), int(c)]

    def is_reference(self):                      # a view of caller memory (featurepatch.cc:45), never an owned copy
        return True

    def has_data(self):
        return True

    def data_ptr(self):
        return self._ptr

    def __setstate__(self, state):               # a copy / unpickled patch owns new memory: the cached address moves with it
        self.__dict__.update(state)
        self._ptr = self.data.ctypes.data
        self._meta = (self._ptr,) + tuple(self._meta[1:])

    def num_bytes(self):                         # Size() * sizeof(dtype)
        return int(self.data.nbytes)

    def current_memory(self):                    # featurepatch.cc CurrentMemory: NumBytes() while data is held
        return self.num_bytes()

    def lock(self):
        pass

    def flush(self):                             # locked / referenced patches free nothing
        return 0

    def get_pixel_coords(self, xy):              # GetPixelCoordinatesVec
        return self.to_pixel_coordinates(np.asarray(xy, dtype=np.float64).reshape(2))

    def to_corner(self, xy, patch_size):         # featurepatch.cc:322-334
        o = patch_size / 2.0
        corner = np.trunc(self.get_pixel_coords(xy) - o).astype(np.int32)          # Eigen cast<int>: towards zero
        corner = np.maximum(corner, 0)
        return np.minimum(corner, np.array([self.width - patch_size, self.height - patch_size], np.int32))

    def slice(self, xy, patch_size):             # featurepatch.cc:336-357: a patch_size window of a (dense) patch, copied
        patch_size = int(patch_size)
        if patch_size > self.width or patch_size > self.height:
            raise ValueError("patch_size exceeds the patch")                    # THROW_CHECK_LE
        c = self.to_corner(xy, patch_size)
        return FeaturePatch(self.data[c[1]:c[1] + patch_size, c[0]:c[0] + patch_size].copy(), c, self.scale)

    def to_pixel_coordinates(self, xy):          # featurepatch.h:250-255
        return np.asarray(xy) * self.scale - 0.5 - self.corner

    def to_image_coordinates(self, uv):          # featurepatch.h:257-262
        return (np.asarray(uv) + self.corner + 0.5) / self.scale


class FeatureMap:
    """Patches of one image keyed by keypoint index (features/src/featuremap.h:104-118)."""

    def __init__(self, patches=None, point2D_ids=None, corners=None, metadata=None, is_sparse=True):
        """FeatureMap({keypoint id: FeaturePatch}, is_sparse=...), or the reference's numpy constructor
        FeatureMap(patches [N][H][W][C], point2D_ids, corners, metadata) (featuremap.cc:8-45; extract.py:131-139): patch i is a
        view of patches[i] at corners[i] with metadata['scale']; a dense map (metadata['is_sparse'] false) holds ONE patch
        under kDenseId."""
        if isinstance(patches, np.ndarray):
            if patches.ndim != 4:
                raise ValueError("patches must be N x H x W x C")                   # THROW_CHECK_EQ(shape.size(), 4)
            self.is_sparse = bool(metadata["is_sparse"])
            if not self.is_sparse and len(patches) != 1:
                raise ValueError("a dense feature map holds exactly one patch")     # THROW_CHECK(is_sparse_ || n_patches == 1)
            scale = np.asarray(metadata["scale"], dtype=np.float64).reshape(2)
            self.patches = {(int(point2D_ids[i]) if self.is_sparse else kDenseId): FeaturePatch(patches[i], corners[i], scale)
                            for i in range(len(patches))}
            self._remember_stack(patches, corners, scale, point2D_ids if self.is_sparse else None)
            return
        self.patches = dict(patches or {})
        self.is_sparse = is_sparse

    _stack = None

    def __getstate__(self):            # pickle / copy.deepcopy: the copy's patches own new memory -- it remembers no stack
        state = dict(self.__dict__)
        state["_stack"] = None
        return state

    class _StackViews(dict):
        """The {keypoint id: FeaturePatch} dict of a map built from ONE stacked array: any write to it -- a patch replaced in
        place, not only through add_fpatch -- drops the owner's remembered stack, so that SharedArena.prefetch can never
        upload a stale slice of the array into the slot of a patch that is no longer a view of it."""
        __slots__ = ("owner",)

        def _dirty(self):
            owner = self.owner() if getattr(self, "owner", None) is not None else None
            if owner is not None:
                owner._stack = None

        # pickle / copy / deepcopy: as a PLAIN dict (a weak reference cannot be pickled, and a copy that kept `owner` would drop
        # the ORIGINAL map's stack on a write to the copy -- ADVICE r5); the copy simply has no remembered stack
        def __reduce__(self):
            return (dict, (dict(self),))

        def __copy__(self):
            return dict(self)

        def __deepcopy__(self, memo):
            import copy
            return {copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()}

        def __setitem__(self, k, v):
            self._dirty(); dict.__setitem__(self, k, v)

        def __delitem__(self, k):
            self._dirty(); dict.__delitem__(self, k)

        def __ior__(self, other):
            self._dirty(); return dict.__ior__(self, other)

        def pop(self, *a):
            self._dirty(); return dict.pop(self, *a)

        def popitem(self):
            self._dirty(); return dict.popitem(self)

        def clear(self):
            self._dirty(); dict.clear(self)

        def update(self, *a, **k):
            self._dirty(); dict.update(self, *a, **k)

        def setdefault(self, *a):
            self._dirty(); return dict.setdefault(self, *a)

    def _remember_stack(self, patches, corners, scale, ids=None):
        """The patches of this map are views of ONE N x H x W x C array (the reference's numpy constructor, featuremap.cc:8-45):
        remembered, so that an upload can take the N patches from the array's address and stride without touching the N
        FeaturePatch objects (SharedArena.prefetch).  Dropped as soon as the dict no longer mirrors the array."""
        if isinstance(patches, np.ndarray) and patches.ndim == 4 and patches.flags["C_CONTIGUOUS"] and len(patches) == len(self.patches) \
                and patches.dtype in (np.float16, np.float32, np.float64) and self.is_sparse:
            views = FeatureMap._StackViews(self.patches)        # same content and order; writes from now on drop the stack
            views.owner = weakref.ref(self)
            self.patches = views
            self._stack = (patches, np.ascontiguousarray(corners, dtype=np.int32).reshape(len(patches), 2),
                           np.asarray(scale, dtype=np.float64).reshape(2),
                           None if ids is None else np.asarray(ids, dtype=np.int64).reshape(len(patches)))   # keypoint id of every row

    def stacked(self):
        """(array, corners, scale) when the dict still holds exactly the views of the array it was built from, else None."""
        st = self._stack
        if st is None or len(self.patches) != len(st[0]) or len(st[0]) == 0:
            return None
        first = next(iter(self.patches.values()))
        if type(first) is not FeaturePatch or first._ptr != st[0].ctypes.data or type(self.patches) is not FeatureMap._StackViews:
            return None          # (`patches` rebound to another dict: nothing watches its writes)
        return st

    @classmethod
    def from_arrays(cls, patches, keypoint_ids, corners, scale):
        """Same inputs as the reference's FeatureMap(patches, point2D_ids, corners, metadata) numpy
        constructor (features/src/featuremap.cc:25-61): N x H x W x C patches."""
        scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), (2,))
        fm = cls({int(k): FeaturePatch(p, c, scale) for k, p, c in zip(keypoint_ids, patches, corners)})
        fm._remember_stack(patches, corners, scale, keypoint_ids)
        return fm

    @classmethod
    def dense(cls, featuremap_hwc, scale):
        """A dense map (is_sparse = False, extractor.py:200-212): ONE patch under kDenseId that covers the whole
        image, corner (0, 0); every keypoint of the image resolves to it."""
        return cls({kDenseId: FeaturePatch(featuremap_hwc, (0, 0), scale)}, is_sparse=False)

    def fpatch(self, keypoint_id):               # featuremap.h:104-111
        return self.patches[int(keypoint_id)] if self.is_sparse else self.patches[kDenseId]

    def has_fpatch(self, keypoint_id):           # featuremap.h:113-118
        if self.is_sparse:
            return int(keypoint_id) in self.patches
        return len(self.patches) == 1 and kDenseId in self.patches

    def keys(self):
        return list(self.patches.keys())

    # -- the rest of the pybind surface (features/bindings.cc:89-113) --
    fpatches = property(lambda self: self.patches)

    def add_fpatch(self, point2D_idx, patch):    # featuremap.h:120-129
        self.patches[int(point2D_idx)] = patch
        self._stack = None

    def num_fpatches(self):
        return len(self.patches)

    @property
    def channels(self):
        for p in self.patches.values():
            return p.shape[2]
        return -1                                # FeatureMap(): channels_(-1)

    def shape(self):                             # featuremap.h:136-156: [n, H, W, C]
        out = [len(self.patches), 0, 0, self.channels]
        for p in self.patches.values():
            out[1] = out[1] or p.shape[0]
            out[2] = out[2] or p.shape[1]
        return out

    @property
    def size(self):
        return sum(p.data.size for p in self.patches.values())

    def num_bytes(self):
        return sum(p.data.nbytes for p in self.patches.values())

    def current_memory(self):
        return self.num_bytes()

    def lock(self):
        pass

    def flush(self):
        return 0


class FeatureSet:
    """One feature level: image name -> FeatureMap (features/src/featureset.h)."""

    def __init__(self, feature_dict=None, channels=None, fmaps=None):
        """FeatureSet(feature_dict, channels) / FeatureSet(channels) like the pybind constructors (features/bindings.cc:127-130)."""
        if isinstance(feature_dict, (int, np.integer)) and channels is None:
            feature_dict, channels = None, int(feature_dict)
        self.fmaps = dict(fmaps if fmaps is not None else (feature_dict or {}))
        self._channels = channels

    def fmap(self, image_name):
        return self.fmaps[image_name]

    def has_fmap(self, image_name):
        return image_name in self.fmaps

    @property
    def channels(self):
        if self._channels is None:
            for fm in self.fmaps.values():
                for p in fm.patches.values():
                    return p.shape[2]
        return self._channels

    # -- the rest of the pybind surface (features/bindings.cc:126-147); the cache bookkeeping of H5-backed sets has nothing
    # to do here: everything is resident --
    def keys(self):
        return list(self.fmaps.keys())

    def add_fmap(self, name, fmap):              # featureset.h AddFeatureMap
        self.fmaps[name] = fmap

    def emplace(self, name, fmap):
        self.fmaps.setdefault(name, fmap)

    def num_bytes(self):
        return sum(fm.num_bytes() for fm in self.fmaps.values())

    def current_memory(self):
        return self.num_bytes()

    def lock(self):
        pass

    def flush(self):
        return 0

    def flush_every_n(self, n):
        pass

    def use_parallel_io(self, do_parallel):
        pass

    def _first_patch(self):
        for fm in self.fmaps.values():
            for p in fm.patches.values():
                return p
        raise ValueError("empty FeatureSet")


class FeatureManager:
    """features/src/featuremanager.h: one FeatureSet per feature level."""

    def __init__(self, fsets, dummy=None, level_prefix=""):
        """FeatureManager([FeatureSet, ...]); or like the pybind constructors (features/bindings.cc:235-237):
        FeatureManager(channels_per_level, dtype_array) -> empty sets to emplace() maps into (extract.py:95-96), and
        FeatureManager(h5_path, fill, level_prefix) -> the cache reader (load_features_from_cache)."""
        if isinstance(fsets, (str, bytes)) or hasattr(fsets, "__fspath__"):
            loaded = load_features_from_cache(fsets, fill=True if dummy is None else bool(dummy), level_prefix=level_prefix)
            fsets = list(loaded.fsets)
        elif len(fsets) and all(isinstance(c, (int, np.integer)) for c in fsets):
            fsets = [FeatureSet(channels=int(c)) for c in fsets]
        self.fsets = _CallableList(fsets)        # `.fsets` here, `.fsets()` in the pybind class: both work

    @property
    def num_levels(self):
        return len(self.fsets)

    def fset(self, level_index):
        return self.fsets[level_index]

    def num_bytes(self):
        return sum(fs.num_bytes() for fs in self.fsets)

    def current_memory(self):
        return self.num_bytes()

    def lock(self):
        pass


class Reference:
    """features/src/references.h:29-72 (N_NODES = 1): source observation + 1 x C descriptor."""

    def __init__(self, image_id=0, point2D_idx=0, descriptor=None, observations=None, costs=None, source=None, track=None):
        """Reference(image_id, point2D_idx, descriptor, observations), or by keyword like the pybind struct (features/bindings.cc:
        264-274; store_references.py:45-53): Reference(descriptor=, observations=, costs=, source=TrackElement, track=Track)."""
        if source is not None:
            image_id, point2D_idx = source.image_id, source.point2D_idx
        self.source = TrackElementTuple(int(image_id), int(point2D_idx))
        self.descriptor = np.zeros((1, 0)) if descriptor is None else np.asarray(descriptor, dtype=np.float64).reshape(1, -1)
        # per-observation descriptors (references.h:52-60), used by "all"-reference localization
        self.observations = [np.asarray(o, dtype=np.float64).reshape(1, -1) for o in (observations if observations is not None else [])]
        # per-observation squared distances to the robust mean, and the visible track they belong to (reference_extractor.h:259-265)
        self.costs = [float(c) for c in (costs if costs is not None else [])]
        self.track = track if track is not None else []

    @property
    def track(self):
        return self._track

    @track.setter
    def track(self, value):
        elems = getattr(value, "elements", value)
        self._track = TrackList(TrackElementTuple(int(e.image_id), int(e.point2D_idx)) if hasattr(e, "image_id")
                                else TrackElementTuple(int(e[0]), int(e[1])) for e in elems)

    channels = property(lambda self: int(self.descriptor.shape[1]))
    n_nodes = property(lambda self: int(self.descriptor.shape[0]))

    def has_observations(self):
        return len(self.observations) > 0


class ReferenceMap(_Mapping):
    """{point3D_id: Reference} as ReferenceExtractor.run returns it, held as three arrays (ids, source observations,
    descriptors) -- a Reference OBJECT is made when somebody asks for one (references[pid], .items(), ...) and kept.  The
    bundle optimiser takes the descriptor rows straight from the array: at BASELINE configs[2] (200k points) building and
    re-reading 200k Python objects cost 2.4 s around a 60 ms solve.  Assignment (references[pid] = Reference(...)) works."""

    def __init__(self, ids, sources, descriptors):
        self._ids = [int(p) for p in ids]
        self._row = {p: k for k, p in enumerate(self._ids)}
        self._src = np.asarray(sources, dtype=np.int64).reshape(len(self._ids), 2)
        d = np.ascontiguousarray(descriptors, dtype=np.float64)
        self._desc = d.reshape(len(self._ids), -1) if len(self._ids) else np.zeros((0, d.shape[-1] if d.ndim == 2 else 0))
        self._objs = {}

    def __len__(self):
        return len(self._row)

    def __iter__(self):
        return iter(self._ids)

    def __contains__(self, pid):
        return pid in self._row

    def __getitem__(self, pid):
        obj = self._objs.get(pid)
        if obj is None:
            k = self._row[pid]                    # KeyError like dict / references.at()
            obj = self._objs[pid] = Reference(int(self._src[k, 0]), int(self._src[k, 1]), self._desc[k])
        return obj

    def __setitem__(self, pid, ref):
        pid = int(pid)
        if pid not in self._row:
            self._row[pid] = len(self._ids)
            self._ids.append(pid)
            self._src = np.concatenate([self._src, [[ref.source[0], ref.source[1]]]])
            self._desc = np.concatenate([self._desc, ref.descriptor.reshape(1, -1)]) if self._desc.size else ref.descriptor.reshape(1, -1).copy()
        self._objs[pid] = ref

    @property
    def channels(self):
        return int(self._desc.shape[1])

    def descriptor_matrix(self, point_ids):
        """(len(point_ids), C) reference descriptors; objects handed out (and possibly edited) or assigned win over the arrays."""
        rows = np.fromiter((self._row[p] for p in point_ids), dtype=np.int64, count=len(point_ids))
        out = self._desc[rows]
        if self._objs:
            at = {p: k for k, p in enumerate(point_ids)}
            for p, ref in self._objs.items():
                if p in at:
                    out[at[p]] = ref.descriptor.reshape(-1)
        return out

    def arrays(self):
        """(ids, sources, descriptors) with the handed-out objects' current values folded in; None if an object carries
        observations / costs / a track (then only the objects describe the map)."""
        if any(r.observations or r.costs or len(r.track) for r in self._objs.values()):
            return None
        desc, src = self._desc.copy(), self._src.copy()
        for p, ref in self._objs.items():
            desc[self._row[p]] = ref.descriptor.reshape(-1)
            src[self._row[p]] = (ref.source[0], ref.source[1])
        return np.asarray(self._ids, dtype=np.int64), src, desc


class ArenaPatch:
    """A patch that already lives in a device arena (written by tensor_to_arena / pxr_arena_extract, or adopted
    from a torch tensor): the FeaturePatch of the GPU-resident flow.  FeatureMap / FeatureSet hold these like
    FeaturePatch objects; the optimisers then index the arena in place instead of stacking and uploading."""

    def __init__(self, arena, index):
        self.arena, self.index = arena, int(index)

    @property
    def shape(self):
        return (self.arena.H, self.arena.W, self.arena.C)


class ArenaRef:
    """What to_arena returns: the arena the kernels read + `index`, the arena patch of each list entry.
    Attribute access falls through to the arena; close() frees it only when to_arena created it."""

    def __init__(self, arena, index, owned):
        self._arena, self.index, self._owned = arena, np.ascontiguousarray(index, dtype=np.int64), owned

    def __getattr__(self, name):
        return getattr(self._arena, name)

    def close(self):
        if self._owned:
            self._arena.close()


_HOST = []


def _host_module():
    """pixsfm_amd._pxr_host (the compiled host binding), or None where it was not built."""
    if not _HOST:
        try:
            from .. import _pxr_host
            _HOST.append(_pxr_host)
        except ImportError:
            _HOST.append(None)
    return _HOST[0]


class SharedArena:
    """One upload for several consumers of the same host patches: BundleAdjuster.refine runs the reference extraction and
    the optimiser on the same FeatureSet -- the second to_arena(..., cache=this) finds its patches in the arena the first
    one built instead of sending 64 KB per observation over PCIe again.  A context manager: the arena is freed on exit."""

    def __init__(self):
        self.arena, self.slot, self.uniq = None, {}, None     # slot: id(patch) -> arena slot (built on demand); uniq: the patches by slot
        self._thread, self._error = None, None
        self.layout, self._lut = None, None      # prefetched stacks: {image name: (first slot, keypoint id of every row)}; its lookup table

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def prefetch(self, ctx, feature_set, image_names):
        """Start uploading every patch of the feature maps of `image_names` in a background thread (the C call that gathers
        into pinned buffers and drives the copy engine holds no GIL), so that the 65 GB of BASELINE configs[2] cross PCIe
        WHILE the caller walks the reconstruction's Python objects -- 1.33 s + 0.54 s back to back in round 3's
        BundleAdjuster.refine.  Only for maps that still mirror the N x H x W x C array they were constructed from
        (FeatureMap.stacked: the reference's numpy constructor, what extract.py hands over): the patch addresses then come
        from the arrays' base and stride, no FeaturePatch object is touched (reading a million cold objects costs as much as
        the walk this is meant to hide: measured, 0.77 s).  Returns False when a map does not qualify; a later
        to_arena(..., cache=self) waits for the upload and, should it need a patch that was not prefetched, uploads its own."""
        import threading
        from ..engine import PatchArena
        if _host_module() is None or self.arena is not None or self._thread is not None:
            return False
        stacks, uniq, layout, base = [], [], {}, 0
        origin = {}                                  # image name -> (the map object, its stack's address): checked again in slots()
        for name in image_names:
            if not feature_set.has_fmap(name):
                continue
            fm = feature_set.fmap(name)
            st = fm.stacked() if hasattr(fm, "stacked") else None
            if st is None:
                return False
            stacks.append(st)
            origin[name] = (fm, int(st[0].ctypes.data))
            uniq.extend(fm.patches.values())         # dict order = array order (insertion order of the constructor)
            ids = st[3] if len(st) > 3 and st[3] is not None else np.fromiter(fm.patches.keys(), dtype=np.int64, count=len(fm.patches))
            layout[name] = (base, ids)
            base += len(st[0])
        if not stacks:
            return False
        shape, dtype = stacks[0][0].shape[1:], stacks[0][0].dtype
        if any(st[0].shape[1:] != shape or st[0].dtype != dtype for st in stacks):
            return False
        pb = int(np.prod(shape)) * dtype.itemsize
        pointers = np.concatenate([np.uint64(st[0].ctypes.data) + np.arange(len(st[0]), dtype=np.uint64) * np.uint64(pb) for st in stacks])
        corners = np.concatenate([st[1] for st in stacks])
        scales = np.concatenate([np.broadcast_to(st[2], (len(st[0]), 2)) for st in stacks])
        self.uniq, self.layout, self._lut = uniq, layout, None
        self._origin = (feature_set, origin)

        def work():
            try:
                self.arena = PatchArena.from_patch_pointers(ctx, pointers, shape, dtype, corners, scales)
            except BaseException as e:  # noqa: BLE001 -- handed to the waiting thread
                self._error = e
        self._thread = threading.Thread(target=work, name="pxr-prefetch", daemon=True)
        self._thread.start()
        return True

    def still_mirrors(self, feature_set):
        """True when `feature_set` is the set this arena was prefetched from and every one of its maps still mirrors the stacked
        array that was uploaded (the layout maps (image name, keypoint id) to a slot -- for another feature set, or after
        a patch was replaced, those slots hold stale data; the caller then goes through the patch objects, which are checked by
        identity)."""
        src = getattr(self, "_origin", None)
        if src is None or feature_set is None or src[0] is not feature_set:
            return False
        for name, (fm, addr) in src[1].items():
            if not feature_set.has_fmap(name) or feature_set.fmap(name) is not fm:
                return False
            st = fm.stacked()
            if st is None or int(st[0].ctypes.data) != addr:
                return False
        return True

    def slots(self, image_names, obs_image, obs_p2d, feature_set=None):
        """Arena slot of every observation (image position `obs_image` into `image_names`, keypoint id `obs_p2d`) straight from the
        prefetched stacks' layout -- first slot of the image's stack + the row of the keypoint -- with numpy only: no FeaturePatch
        object is looked up or touched (scene.patches_of + slots_of read a million of them: 0.3 + 0.1 s at BASELINE configs[2]).
        None when the arena does not come from a prefetch or does not hold every observation's patch (the caller then goes through
        the patch objects)."""
        self.wait()
        if self.arena is None or not self.layout:
            return None
        if not self.still_mirrors(feature_set):
            return None
        if self._lut is None or self._lut[0] != tuple(image_names):
            ptr, parts = np.zeros(len(image_names) + 1, np.int64), []
            for k, name in enumerate(image_names):
                ent = self.layout.get(name)
                if ent is None or len(ent[1]) == 0:
                    ptr[k + 1] = ptr[k]
                    continue
                base, ids = ent
                if ids.min() < 0 or ids.max() > 8 * len(ids) + 65536:          # sparse ids: no table (the object path handles them)
                    return None
                lut = np.full(int(ids.max()) + 1, -1, np.int64)
                lut[ids] = base + np.arange(len(ids), dtype=np.int64)
                parts.append(lut)
                ptr[k + 1] = ptr[k] + len(lut)
            self._lut = (tuple(image_names), ptr, np.concatenate(parts) if parts else np.zeros(0, np.int64))
        _, ptr, table = self._lut
        oi, oj = np.asarray(obs_image, dtype=np.int64), np.asarray(obs_p2d, dtype=np.int64)
        if len(oi) == 0 or len(table) == 0:
            return None
        size = ptr[oi + 1] - ptr[oi]
        ok = (oj >= 0) & (oj < size)
        if not ok.all():
            return None
        idx = table[ptr[oi] + oj]
        return idx if (idx >= 0).all() else None

    def wait(self):
        """Join a running prefetch; its failure (e.g. out of device memory) leaves the cache empty and the caller uploads."""
        t, self._thread = self._thread, None
        if t is not None:
            t.join()
            if self._error is not None or self.arena is None:
                if self._error is not None and not isinstance(self._error, Exception):
                    raise self._error                      # KeyboardInterrupt / SystemExit are not the upload's to swallow
                if self._error is not None:
                    warnings.warn("pixsfm_amd: the background upload of the stacked feature maps failed (%r); "
                                  "uploading patch by patch instead" % (self._error,), RuntimeWarning, stacklevel=2)
                self._error, self.arena, self.uniq, self.layout, self._lut = None, None, None, None, None

    def close(self):
        self.wait()
        if self.arena is not None:
            self.arena.close()
        self.arena, self.slot, self.uniq, self.layout, self._lut = None, {}, None, None, None


def to_arena(ctx, patch_list, cache=None):
    """FeaturePatch objects (identical H, W, C, dtype) are stacked into a new HBM arena; ArenaPatch objects
    of one arena are used where they are (no copy).  cache: a SharedArena that keeps / provides the upload."""
    from ..engine import PatchArena
    if not patch_list:
        raise ValueError("no patches")
    if cache is not None:
        cache.wait()                             # a prefetch started by the adjuster (SharedArena.prefetch)
    if cache is not None and cache.arena is not None:
        host = _host_module()
        if host is not None and hasattr(host, "slots_of") and cache.uniq is not None:
            index = host.slots_of(cache.uniq, list(patch_list))            # identity lookup in C++ (no dict of a million ids)
            if len(index) and index.min() >= 0:
                return ArenaRef(cache.arena, index, owned=False)
            cache = None                         # a patch the shared arena does not hold: an arena of this call's own
        else:
            if not cache.slot and cache.uniq is not None:
                cache.slot = {id(p): k for k, p in enumerate(cache.uniq)}
            try:
                return ArenaRef(cache.arena, [cache.slot[id(p)] for p in patch_list], owned=False)
            except KeyError:
                cache = None
    on_device = [isinstance(p, ArenaPatch) for p in patch_list]
    if all(on_device):
        arena = patch_list[0].arena
        if any(p.arena is not arena for p in patch_list):
            raise ValueError("the patches of one problem must live in one arena")
        return ArenaRef(arena, [p.index for p in patch_list], owned=False)
    if any(on_device):
        raise ValueError("cannot mix host FeaturePatch and device ArenaPatch objects in one problem")
    # one arena entry per DISTINCT patch object: in dense mode all keypoints of an image share one (large) patch
    host = _host_module()
    mixed = "the accelerated path needs patches of identical shape and dtype (sparse patches of one patch_size, " \
            "pixsfm/features/extractor.py:33-51, or dense maps of equal size)"
    if host is not None:             # compiled walk over the patch objects (csrc/pybind/pxr_host.cpp)
        try:
            index, uniq, pointers, corners, scales = host.gather_patches(list(patch_list))
        except ValueError:
            raise ValueError(mixed) from None
        slot = None
    else:
        slot, uniq, index = {}, [], np.empty(len(patch_list), dtype=np.int64)
        for k, p in enumerate(patch_list):
            if id(p) not in slot:
                slot[id(p)] = len(uniq)
                uniq.append(p)
            index[k] = slot[id(p)]
        corners = np.array([p.corner for p in uniq], dtype=np.int32).reshape(len(uniq), 2)
        scales = np.array([p.scale for p in uniq], dtype=np.float64).reshape(len(uniq), 2)
        pointers = np.fromiter((p.data_ptr() for p in uniq), dtype=np.uint64, count=len(uniq))
        if len({p._meta[5] for p in uniq}) != 1:
            raise ValueError(mixed)
    shape, dtype = uniq[0].shape, uniq[0].data.dtype
    # the patches go up one by one through pinned staging buffers (no np.stack of the set: 65 GB at BASELINE configs[2])
    arena = PatchArena.from_patch_pointers(ctx, pointers, shape, dtype, corners, scales)
    if cache is not None:
        cache.arena, cache.uniq = arena, uniq
        cache.slot = slot if slot is not None else {}        # the compiled path looks patches up by identity in C++ (slots_of)
        return ArenaRef(arena, index, owned=False)
    return ArenaRef(arena, index, owned=True)


def tensor_to_arena(arena, first, featuremap, image_size, keypoints, l2_normalize=True):
    """GPU-resident counterpart of FeatureExtractor.tensor_to_fmap(featuremap, image_size, keypoints)
    (pixsfm/features/extractor.py:152-199, sparse branch): the dense map stays on the device and its
    16 x 16 windows are written straight into `arena` (patches first .. first + n - 1, with corners and
    scale) by pxr_arena_extract -- no numpy copy, no PCIe crossing (extract_patches.py:41-44 names that
    copy the main bottleneck).  featuremap: torch.cuda tensor (1, C, h, w) or (C, h, w), fp16/fp32,
    contiguous; image_size: (width, height); returns the number of patches written."""
    return arena.extract(first, featuremap, keypoints, image_size, l2_normalize=l2_normalize)


def fmap_from_arena(arena, first, keypoint_ids):
    """FeatureMap over patches first .. first + len(keypoint_ids) - 1 of a device arena (e.g. the ones
    tensor_to_arena just wrote for one image)."""
    fm = FeatureMap()
    for k, kid in enumerate(keypoint_ids):
        fm.patches[int(kid)] = ArenaPatch(arena, first + k)
    return fm


class PatchInterpolator:
    """_features.PatchInterpolator(interpolation_config) (features/bindings.cc; features/src/patch_interpolator.h):
    interpolate_nodes(fpatch, xy) -> (1, C) descriptor of the patch at a keypoint in COLMAP image coordinates, on
    the GPU (pxr_interpolate).  interpolate_many is the batched form the kernels are made for."""

    def __init__(self, interpolation_config=None, ctx=None):
        from . import base
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx = ctx

    def interpolate_many(self, fpatches, xys, jacobian=False):
        from ..engine import interpolate
        from .keypoint_adjustment import default_context
        ctx = self.ctx or default_context()
        arena = to_arena(ctx, list(fpatches))
        desc, J = interpolate(ctx, arena, self.interpolation.to_engine(), xys, arena.index, jacobian=jacobian)
        arena.close()
        return (desc, J) if jacobian else desc

    def interpolate_nodes(self, fpatch, xy):
        return self.interpolate_many([fpatch], np.asarray(xy, dtype=np.float64).reshape(1, 2))

    def interpolate(self, fpatch, xy):
        return self.interpolate_nodes(fpatch, xy).reshape(-1)

    def interpolate_local(self, fpatch, xy):
        """InterpolateLocal (dynamic_patch_interpolator.h:125-132): `xy` in the patch's own pixel coordinates (column, row), no
        image -> patch transform."""
        local = FeaturePatch(fpatch.data, (0, 0), (1.0, 1.0))
        return self.interpolate(local, np.asarray(xy, dtype=np.float64).reshape(2) + 0.5)


def load_features_from_cache(cache_path, fill=True, level_prefix="", ctx=None, device=False, required=None):
    """pixsfm.extract.load_features_from_cache(cache_path, fill) (extract.py:218-222) -> FeatureManager, through the native
    reader of the cache format (libpixsfm_h5.so: featuremanager.cc:20-40, featuremap.cc:60-267; "chunked" and "grouped"
    files, dense maps stored once and loaded as sparse windows).
    device=False: host FeaturePatch objects (numpy arrays read straight from the file).
    device=True: every level becomes ONE device PatchArena (image by image: read into a staging array, upload) and the
    FeatureMaps hold ArenaPatch handles -- the flow the optimisers then index in place.
    required: optional {image name: iterable of keypoint ids}: only these patches are read (FeatureSet::Load with the
    patch ids a FeatureView needs, featureset.cc:98-135); images not named are skipped.
    fill=False (metadata now, data on demand) is expressed with `required` here."""
    from .. import _h5
    from ..engine import PatchArena
    if not fill:
        raise ValueError("fill=False is not supported: pass required={image: keypoint ids} to load a subset")
    cache = _h5.FeatureCache(cache_path, level_prefix)
    fsets = []
    try:
        for level in range(cache.num_levels):
            names = [n for n in cache.image_names(level) if required is None or n in required]
            plan, total, shape = [], 0, None
            for name in names:
                info = cache.map_info(level, name)
                ids, corners, scales = cache.map_meta(level, name, info["n"])
                which = np.arange(info["n"])
                if required is not None and info["is_sparse"]:
                    pos = {int(k): i for i, k in enumerate(ids)}
                    missing = [k for k in required[name] if int(k) not in pos]
                    if missing:
                        raise KeyError("keypoints %r of image %r are not in the cache" % (missing[:5], name))
                    which = np.array([pos[int(k)] for k in required[name]], dtype=np.int64)
                plan.append((name, info, ids, corners, scales, which))
                total += len(which)
                if device and len(which):
                    if shape is None:
                        shape = info["shape"]
                    elif shape != info["shape"]:
                        raise ValueError("device=True needs patches of one shape per level (image %r has %r, expected %r)"
                                         % (name, info["shape"], shape))
            fset = FeatureSet(channels=cache.channels_per_level[level])
            arena, first = None, 0
            if device and total:
                from .keypoint_adjustment import default_context
                arena = PatchArena(ctx or default_context(), total, shape[0], shape[1], shape[2], cache.dtype)
                fset.arena = arena
            for name, info, ids, corners, scales, which in plan:
                patches = cache.read_patches(level, name, which)
                fm = FeatureMap(is_sparse=info["is_sparse"])
                for j, i in enumerate(which):
                    key = int(ids[i]) if info["is_sparse"] else kDenseId
                    fm.patches[key] = ArenaPatch(arena, first + j) if arena is not None else \
                        FeaturePatch(patches[j], corners[i], scales[i])
                if arena is not None and len(which):
                    arena.upload(first, patches, corners[which], scales[which])
                    first += len(which)
                fset.fmaps[name] = fm
            fsets.append(fset)
    finally:
        cache.close()
    return FeatureManager(fsets)
