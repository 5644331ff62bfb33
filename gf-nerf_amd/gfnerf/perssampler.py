"""PersSampler (gfnerf/perssampler.py:48-135,350-530) on the HIP kernels.

`_NativePersSampler` mirrors the TorchScript class my_classes.PersSampler (bindings.cpp:18-299,359-398): InitSampler
with the reference's 18 positional arguments, GetSamples -> the 8 dense tensors of PersSampler_cuda.cu:467-476,
UpdateOctNodes / UpdateRayMarch / UpdateMode / UpdateBlockIdxs, States / LoadStates with the reference's blobs, and
the get_*_ accessors.  Internally samples stay packed; the dense tensors are an export for callers that want the
reference's layout.  `PersSampler` is the nn.Module wrapper (gfnerf/perssampler.py) with the reference's methods, buffers and
state-dict keys, the block routing (camera clustering, nearest-split lookup) included."""
import math
from typing import Any, List, Mapping

import numpy as np
import torch
from torch import nn

from .. import ops
from .rays import RayBundle, RaySamples, WarpedSamples

TRAIN, VALIDATE = 0, 1
INIT_NODE_STAT = 1000
_NODE_DT = np.dtype({"names": ["center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx", "block_idx"],
                     "formats": [("<f4", 3), "<f4", "<i8", ("<i8", 8), "u1", "<i8", "<i8"],
                     "offsets": [0, 12, 16, 24, 88, 96, 104], "itemsize": 128})
_TRANS_DT = np.dtype({"names": ["w2xz", "weight", "center", "side_len", "dis_summary"],
                      "formats": [("<f4", (12, 2, 4)), ("<f4", (3, 12)), ("<f4", 3), "<f4", "<f4"],
                      "offsets": [0, 384, 528, 540, 544], "itemsize": 576})


class _NativePersSampler:
    def __init__(self):
        self.initialised = False

    def InitSampler(self, split_dist_thres, sub_div_milestones, compact_freq, max_oct_intersect_per_ray, global_near,
                    scale_by_dis, bbox_levels, sample_l, max_level, c2w, w2c, intri, bounds, mode, sampled_oct_per_ray,
                    ray_march_fineness, ray_march_init_fineness, ray_march_fineness_decay_end_iter, seed=1234):
        """PersSampler::PersSampler (PersSampler.cpp:899-952).  w2c is accepted for signature parity and recomputed
        inside the builder."""
        dev = c2w.device
        self.device = dev
        self.sub_div_milestones_ = list(reversed([int(v) for v in sub_div_milestones]))  # :915
        self.compact_freq_ = int(compact_freq)
        self.max_oct_intersect_per_ray_ = int(max_oct_intersect_per_ray)
        self.global_near_ = float(np.float32(global_near))
        self.scale_by_dis_ = bool(scale_by_dis)
        self.sample_l_ = float(np.float32(sample_l))
        self.mode_ = int(mode)
        self.sampled_oct_per_ray_ = float(sampled_oct_per_ray)
        self.ray_march_fineness_ = float(ray_march_fineness)
        self.ray_march_init_fineness_ = float(ray_march_init_fineness)
        self.ray_march_fineness_decay_end_iter_ = int(ray_march_fineness_decay_end_iter)
        bbox_side_len = float(1 << (int(bbox_levels) - 1))  # :921
        nodes, transes = ops.octree_build(c2w.detach().cpu().numpy(), intri.detach().cpu().numpy(),
                                          bounds.detach().cpu().numpy(), max_depth=int(max_level), bbox_side_len=bbox_side_len,
                                          split_dist_thres=float(split_dist_thres), seed=seed)
        # kept for MarkInvisibleNodes (PersOctree keeps intri_, w2c_, bound_ on the device, PersSampler.cpp:100-107)
        self.intri_ = intri.detach().to(dev).float().contiguous()
        self.w2c_ = w2c.detach().to(dev).float().contiguous()
        self.bound_ = bounds.detach().to(dev).float().contiguous()
        self._set_tree(torch.from_numpy(nodes).to(dev), torch.from_numpy(transes).to(dev))
        self.n_volumes_ = self.pers_trans_gpu_.numel() // ops.TRANS_INFO_BYTES  # :936
        self.tree_visit_cnt_ = torch.zeros(self.n_nodes, dtype=torch.int64, device=dev)
        self.generator = torch.Generator(device=dev)
        self.generator.manual_seed(seed)
        self.last_packed = None
        # ray data parallelism: set to gfnerf_amd.dist.TrainSync.reduce_octree_stats and K5's per-node maxima are MAX-reduced
        # over the ranks before the statistics update + K6 (every replica then keeps the same tree)
        self.reduce_max = None
        self.initialised = True

    def _set_tree(self, nodes, transes):
        self.tree_nodes_gpu_ = nodes.contiguous()
        self.pers_trans_gpu_ = transes.contiguous()
        self.n_nodes = self.tree_nodes_gpu_.numel() // ops.TREE_NODE_BYTES
        self.nodes_compact_ = ops.tree_compact(self.tree_nodes_gpu_)
        dev = nodes.device
        self.tree_weight_stats_ = torch.full((self.n_nodes,), INIT_NODE_STAT, dtype=torch.int64, device=dev)
        self.tree_alpha_stats_ = torch.full((self.n_nodes,), INIT_NODE_STAT, dtype=torch.int64, device=dev)
        self._ws = None

    # ---------------------------------------------------------------- sampling
    def make_noise(self, n_rays):
        """rays_noise (PersSampler_cuda.cu:380-389)."""
        n = ops.MAX_SAMPLE_PER_RAY + n_rays + 10
        if self.mode_ == VALIDATE:
            noise = torch.ones(n, dtype=torch.float32, device=self.device)
        else:
            noise = (torch.rand(n, dtype=torch.float32, device=self.device, generator=self.generator) - 0.5) + 1.0
        return noise.mul_(self.ray_march_fineness_)

    def sample_packed(self, rays_o, rays_d, noise=None, table_space=False, want_world=True):
        d = ops.normalize_dirs(rays_d.contiguous().float())
        if noise is None:
            noise = self.make_noise(rays_o.shape[0])
        s = ops.sample_rays(rays_o.contiguous().float(), d, noise, self.nodes_compact_, self.pers_trans_gpu_, self.sample_l_,
                            self.scale_by_dis_, near=self.global_near_, far=1e8, max_hits=self.max_oct_intersect_per_ray_,
                            want_world=want_world, want_dirs=want_world, table_space=table_space, workspace=self._ws)
        self._ws = s["workspace"]
        s["dirs_unit"] = d
        R = rays_o.shape[0]
        s["n_dev"] = s["pts_offsets"][R:R + 1]
        if self.mode_ != VALIDATE and R > 0:  # :386-387 running mean of leaves per ray (kept on the device: no sync)
            per_ray = s["workspace"]["hit_counts"][:R].float().mean()
            self.sampled_oct_per_ray_ = self.sampled_oct_per_ray_ * 0.9 + per_ray * 0.1
        self.last_packed = s
        return s

    def GetSamples(self, rays_o, rays_d, bounds) -> List[torch.Tensor]:
        """-> [world_pts, warped_pts, dirs, dists, t, anchors(i64), pts_idx_start_end, first_oct_dis], dense [R,1024,.]
        (bounds is ignored by the reference too, PersSampler_cuda.cu:326-327)."""
        s = self.sample_packed(rays_o, rays_d, table_space=False, want_world=True)
        return ops.samples_to_dense(s, rays_o.shape[0])

    # ------------------------------------------------------------- tree updates
    def UpdateOctNodes(self, sampled_anchors, pts_idx_bounds, sampled_weights, sampled_alpha, iter_step):
        """K5 + statistics + K6 (PersSampler_cuda.cu:584-665).  Accepts the reference's dense [R,1024,.] tensors
        (gathered to the packed order of the last GetSamples) or already-packed [N] weights / alphas."""
        s = self.last_packed
        assert s is not None, "UpdateOctNodes needs the samples of the preceding GetSamples call"
        cap = s["ts"].shape[0]
        if sampled_weights.dim() >= 2 and sampled_weights.shape[1] == ops.MAX_SAMPLE_PER_RAY:
            R = sampled_weights.shape[0]
            counts = (s["pts_offsets"][1:] - s["pts_offsets"][:-1]).long()
            mask = torch.arange(ops.MAX_SAMPLE_PER_RAY, device=self.device)[None, :] < counts[:, None]
            w = torch.zeros(cap, dtype=torch.float32, device=self.device)
            a = torch.zeros(cap, dtype=torch.float32, device=self.device)
            wv = sampled_weights.reshape(R, -1)[mask]
            w[: wv.numel()] = wv
            a[: wv.numel()] = sampled_alpha.reshape(R, -1)[mask]
        else:
            w, a = sampled_weights.contiguous().float(), sampled_alpha.contiguous().float()
        ops.update_nodes(s["pts_offsets"], s["anchors"], w, a, self.tree_weight_stats_, self.tree_alpha_stats_,
                         self.tree_visit_cnt_, self.tree_nodes_gpu_, reduce_max=self.reduce_max)
        # subdivision milestones and the periodic compaction (:667-676)
        changed = False
        while self.sub_div_milestones_ and self.sub_div_milestones_[-1] <= iter_step:
            self.ProcOctree(True, True, self.sub_div_milestones_[-1] <= 0)
            self.MarkInvisibleNodes()
            self.ProcOctree(True, False, False)
            self.sub_div_milestones_.pop()
            changed = True
        if iter_step % self.compact_freq_ == 0:
            self.ProcOctree(True, False, False)
            changed = True
        if not changed:
            ops.tree_compact(self.tree_nodes_gpu_, self.nodes_compact_)  # K6 may have invalidated leaves

    def ProcOctree(self, compact: bool, subdivide: bool, brute_force: bool):
        """PersOctree::ProcOctree (PersSampler.cpp:154-417) on the device (gfn_octree_proc): the reference round-trips the blob and
        the statistics through the host here; only the new node count is read back.  Visit counts restart at zero (:415)."""
        if not compact:
            raise ops._lib.GfnError("ProcOctree: compact = False is not a state the reference can be in (PersSampler_cuda.cu:668-675)")
        self.tree_nodes_gpu_, self.tree_weight_stats_, self.tree_alpha_stats_ = ops.octree_proc(
            self.tree_nodes_gpu_, self.tree_weight_stats_, self.tree_alpha_stats_, self.tree_visit_cnt_, subdivide=subdivide, brute_force=brute_force)
        self.n_nodes = self.tree_nodes_gpu_.numel() // ops.TREE_NODE_BYTES
        self.tree_visit_cnt_ = torch.zeros(self.n_nodes, dtype=torch.int64, device=self.device)
        self.nodes_compact_ = ops.tree_compact(self.tree_nodes_gpu_)
        self._ws = None
        self.last_packed = None  # node indices of the previous samples are stale now

    def MarkInvisibleNodes(self):
        """K7 (PersSampler_cuda.cu:709-742)."""
        ops.mark_invisible_nodes(self.tree_nodes_gpu_, self.intri_, self.w2c_, self.bound_)
        ops.tree_compact(self.tree_nodes_gpu_, self.nodes_compact_)

    def VisOctree(self, base_exp_dir_):
        """PersSampler::VisOctree (PersSampler.cpp:478-514, bindings.cpp:53,363): the eight corners of every node as `v` lines and the
        twelve edges of every LEAF as `l` lines of <base_exp_dir_>/octree.obj (a debugging aid: host file output, no kernel)."""
        import os

        nd = self._nodes_np()
        with open(os.path.join(base_exp_dir_, "octree.obj"), "w") as f:
            for n in nd:
                c, side = n["center"].astype(np.float32), np.float32(n["side_len"])
                for st in range(8):
                    x, y, z = (c + np.array([(st >> 2 & 1) - 0.5, (st >> 1 & 1) - 0.5, (st & 1) - 0.5], np.float32) * side).tolist()
                    f.write(f"v {x:g} {y:g} {z:g}\n")  # (std::ofstream's default float format: 6 significant digits)
            for i, n in enumerate(nd):
                if not n["is_leaf_node"]:
                    continue
                for a in range(8):
                    for b in range(a + 1, 8):
                        if (a ^ b) in (1, 2, 4):
                            f.write(f"l {i * 8 + a + 1} {i * 8 + b + 1}\n")

    def GetEdgeSamples(self, n_pts):
        """bindings.cpp:84-87,370 -> K9 GetEdgeSamplesKernel (PersSampler_cuda.cu:479-516): random points on the edges shared by neighbouring
        warps, the input of F2-NeRF's total-variation loss.  Its only Python caller, PersSampler.get_edge_samples (perssampler.py:495-497),
        is itself never called anywhere in gfnerf/ (no TV loss in GFNeRFModel.get_loss_dict, nerfacto.py:628-689), and the edge pool it reads
        is never built by GF-NeRF's tree construction: outside the hot path -- refused loudly rather than returning made-up points."""
        raise NotImplementedError("my_classes.PersSampler.GetEdgeSamples (K9, F2-NeRF's TV-loss edge samples) has no caller in gfnerf/ and no edge pool "
                                  "to read: not part of the GF-NeRF hot path (SURVEY.md 2b K9)")

    def get_points_anchors(self, rays_o, rays_d, t_starts, t_ends):
        """bindings.cpp:391 -> K10 GetRaysTreeNodesIntersects / GetTreeNodeIdxFromTs: serves the proposal sampler, which gfnerf/ never creates."""
        raise NotImplementedError("my_classes.PersSampler.get_points_anchors (K10) serves the proposal sampler, which the reference never creates "
                                  "(no `proposal_sampler` assignment anywhere in gfnerf/): dead code, not built (SURVEY.md 2b)")

    def trans_query_frame(self, world_positions_flat, anchors_flat):
        """bindings.cpp:392 -> K11 TransQueryFrameKernel: the same dead proposal-sampler path; samples leave GetSamples warped already."""
        raise NotImplementedError("my_classes.PersSampler.trans_query_frame (K11) serves the dead proposal-sampler path (SURVEY.md 2b)")

    def UpdateRayMarch(self, cur_step):
        if cur_step >= self.ray_march_fineness_decay_end_iter_:  # PersSampler.cpp:958-967
            self.ray_march_fineness_ = 1.0
        else:
            p = float(cur_step) / float(self.ray_march_fineness_decay_end_iter_)
            self.ray_march_fineness_ = math.exp(math.log(1.0) * p + math.log(self.ray_march_init_fineness_) * (1.0 - p))

    def UpdateMode(self, mode):
        self.mode_ = int(mode)

    def UpdateBlockIdxs(self, centers):
        """SetBlockIdxsNearestKernel (PersSampler_cuda.cu:746-766): nearest block centre per node.  One-shot at the
        stage switch; done with tensor ops on views of the blob."""
        nodes = self.tree_nodes_gpu_.view(self.n_nodes, ops.TREE_NODE_BYTES)
        c = nodes[:, :12].contiguous().view(torch.float32).view(self.n_nodes, 3)
        dist = (c[:, None, :].double() - centers[None].double().to(c.device)).norm(dim=-1)
        idx = dist.argmin(dim=1)
        nodes[:, 104:112] = idx.view(-1, 1).contiguous().view(torch.uint8).view(self.n_nodes, 8)
        ops.tree_compact(self.tree_nodes_gpu_, self.nodes_compact_)

    def qurey_tree_nodes_centers(self, anchors):
        """PersSampler::QueryTreeNodeCenters (K12, PersSampler_cuda.cu:984-1029; the binding's spelling, bindings.cpp:396):
        anchors i64 [n,1] node indices -> centers f32 [n,3] (zeros where out of range)."""
        return ops.query_tree_nodes_centers(self.tree_nodes_gpu_, anchors.to(self.device).contiguous())

    # ------------------------------------------------------------------ state
    def States(self) -> List[torch.Tensor]:
        ms = torch.tensor(self.sub_div_milestones_, dtype=torch.int64, device=self.device)
        return [self.tree_nodes_gpu_, self.pers_trans_gpu_, self.tree_visit_cnt_, ms]  # PersSampler.cpp:969-979

    def LoadStates(self, states: List[torch.Tensor], idx: int) -> int:
        dev = self.device
        nodes = states[idx].clone().to(dev); idx += 1
        trans = states[idx].clone().to(dev); idx += 1
        self._set_tree(nodes, trans)
        self.tree_visit_cnt_ = states[idx].clone().to(dev).contiguous(); idx += 1
        self.sub_div_milestones_ = [int(v) for v in states[idx].cpu().tolist()]; idx += 1
        return idx

    # --------------------------------------------------------------- accessors
    def _nodes_np(self):
        return self.tree_nodes_gpu_.cpu().numpy().view(_NODE_DT)

    def get_tree_nodes_center_(self):
        return self._nodes_np()["center"].astype(np.float64).tolist()

    def get_tree_nodes_block_idx_(self):
        return self._nodes_np()["block_idx"].tolist()

    def get_tree_nodes_side_len_(self):
        return self._nodes_np()["side_len"].astype(np.float64).tolist()

    def get_tree_nodes_trans_idx_(self):
        return self._nodes_np()["trans_idx"].tolist()

    def get_tree_nodes_is_leaf_node_(self):
        return [bool(v) for v in self._nodes_np()["is_leaf_node"]]

    def get_pers_trans_info(self):
        t = self.pers_trans_gpu_.cpu().numpy().view(_TRANS_DT)
        return (t["w2xz"].astype(np.float64).tolist(), t["weight"].astype(np.float64).tolist(),
                t["center"].astype(np.float64).tolist(), t["side_len"].astype(np.float64).tolist(),
                t["dis_summary"].astype(np.float64).tolist())

    def get_sub_div_milestones_(self): return list(self.sub_div_milestones_)
    def get_compact_freq_(self): return self.compact_freq_
    def get_max_oct_intersect_per_ray_(self): return self.max_oct_intersect_per_ray_
    def get_global_near_(self): return self.global_near_
    def get_sample_l_(self): return self.sample_l_
    def get_scale_by_dis_(self): return float(self.scale_by_dis_)
    def get_mode_(self): return self.mode_
    def get_n_volumes_(self): return self.n_volumes_
    def get_sampled_oct_per_ray_(self): return float(self.sampled_oct_per_ray_)
    def get_ray_march_fineness_(self): return self.ray_march_fineness_


class _Cameras:
    """What PersSampler needs from nerfstudio's Cameras: camera_to_worlds [n,3,4] + intrinsics matrices [n,3,3]."""

    def __init__(self, camera_to_worlds, intrinsics):
        self.camera_to_worlds = camera_to_worlds
        self._K = intrinsics

    def get_intrinsics_matrices(self):
        return self._K

    def to(self, device):
        return _Cameras(self.camera_to_worlds.to(device), self._K.to(device))

    @property
    def device(self):
        return self.camera_to_worlds.device

    def __len__(self):
        return self.camera_to_worlds.shape[0]


class TreeNodes:
    """perssampler.py:33-45: the per-node columns of the octree as tensors (what `tree_nodes_` returns)."""

    def __init__(self, center, side_len, block_idx, trans_idx, is_leaf_node):
        self.center = center
        self.side_len = side_len
        self.block_idx = block_idx
        self.trans_idx = trans_idx
        self.is_leaf_node = is_leaf_node


class PersSampler(nn.Module):
    def __init__(self, cameras, n_split_dataset, steps_per_split_dataset, steps_perssampler_init, bounds,
                 split_dist_thres: float = 1.5, sub_div_milestones=(2000, 4000, 6000, 8000, 10000), compact_freq: int = 1000,
                 max_oct_intersect_per_ray: int = 1024, global_near: float = 0.01, scale_by_dis: bool = True,
                 bbox_levels: int = 8, sample_l: float = 1.0 / 256, max_level: int = 16, mode: int = 0,
                 sampled_oct_per_ray: int = 512, ray_march_fineness: float = 1.0, ray_march_init_fineness=16.0,
                 ray_march_fineness_decay_end_iter=10000, device=None, seed=1234) -> None:
        super().__init__()
        self.device = torch.device(device or "cuda:0")
        self.cameras = cameras
        c2w = cameras.camera_to_worlds.to(self.device).float()
        n = c2w.shape[0]
        w2c = torch.eye(4, device=self.device).unsqueeze(0).repeat(n, 1, 1).contiguous()
        w2c[:, 0:3, :] = c2w
        w2c = torch.linalg.inv(w2c)[:, 0:3, :].contiguous()
        intri = cameras.get_intrinsics_matrices().to(self.device).float()
        self.bounds = bounds.to(self.device).float()
        self.max_pts_per_ray = 1024
        scale = max(steps_perssampler_init // 30000, 1)  # perssampler.py:99-100
        ray_march_fineness_decay_end_iter = int(ray_march_fineness_decay_end_iter * scale)
        sub_div_milestones = [int(x * scale) for x in sub_div_milestones]
        self.sampler = _NativePersSampler()
        self.sampler.InitSampler(split_dist_thres, sub_div_milestones, compact_freq, max_oct_intersect_per_ray, global_near,
                                 scale_by_dis, bbox_levels, sample_l, max_level, c2w, w2c, intri, self.bounds, mode,
                                 sampled_oct_per_ray, ray_march_fineness, ray_march_init_fineness,
                                 ray_march_fineness_decay_end_iter, seed=seed)
        self._register_state_dict_hook(self.state_dict_hook)
        for k in ("tree_nodes_gpu", "pers_trans_gpu", "tree_visit_cnt", "milestones_ts", "cameras_rays_nodes_centers", "cameras_labels",
                  "cameras_valid_mask"):
            self.register_buffer(k, None)
        self.register_buffer("c2w", c2w)
        self.n_split_dataset = n_split_dataset
        self.steps_per_split_dataset = steps_per_split_dataset
        self.steps_perssampler_init = steps_perssampler_init

    def forward(self, ray_bundle: RayBundle) -> RaySamples:
        return self.generate_ray_samples(ray_bundle)

    def generate_ray_samples(self, ray_bundle: RayBundle, dense: bool = True) -> RaySamples:
        """perssampler.py:350-447.  The step arrives as a Python int / 0-d tensor; no .item() round trips beyond that."""
        cur_step, cur_split_idx = -1, -1
        if ray_bundle.steps is not None:
            cur_step = int(ray_bundle.steps.reshape(-1)[0]) if isinstance(ray_bundle.steps, torch.Tensor) else int(ray_bundle.steps)
            if cur_step >= self.steps_perssampler_init:
                cur_split_idx = ((cur_step - self.steps_perssampler_init) // self.steps_per_split_dataset) % self.n_split_dataset
        packed = self.sampler.sample_packed(ray_bundle.origins, ray_bundle.directions, table_space=False, want_world=dense)
        f2 = None
        if dense:
            R = len(ray_bundle)
            (world, warped, dirs, dists, t, anchors, se, fod) = ops.samples_to_dense(packed, R)
            f2 = WarpedSamples(sampled_world_pts=world, sampled_pts=warped, sampled_dirs=dirs, sampled_dists=dists.unsqueeze(-1),
                               sampled_t=t.unsqueeze(-1), sampled_anchors=anchors,
                               pts_idx_start_end=se.unsqueeze(1).expand(-1, self.max_pts_per_ray, -1),
                               first_oct_dis=fod.unsqueeze(1).expand(-1, self.max_pts_per_ray, -1))
        return RaySamples(f2samples=f2, camera_indices=ray_bundle.camera_indices, rel_camera_indices=ray_bundle.rel_camera_indices,
                          deltas=None if f2 is None else f2.sampled_dists, cur_step=cur_step,
                          cur_split_dataset_idx=cur_split_idx, packed=packed)

    # ------------------------------------------------------------------ block routing (perssampler.py:138-260)
    def get_nearest_split_dataset(self, origin, direction=None):
        """perssampler.py:138-165: the split (= focal block) of the training camera nearest to `origin` -> (split, camera)."""
        cameras_pos = self.c2w[:, 0:3, -1]
        dists = torch.norm(cameras_pos - origin.to(cameras_pos).view(1, 3), dim=1, p=2)
        nearest_image_idx = int(torch.argmin(dists).item())
        return int(self.cameras_labels[nearest_image_idx].item()), nearest_image_idx

    def get_nearest_split_dataset_orig(self, origin):
        """perssampler.py:244-260: before the cameras are clustered, splits are consecutive runs of n_cameras // n_split images."""
        n_cameras = self.c2w.shape[0]
        per = max(n_cameras // self.n_split_dataset, 1)
        cameras_pos = self.c2w[:, 0:3, -1]
        nearest_image_idx = int(torch.argmin(torch.norm(cameras_pos - origin.to(cameras_pos).view(1, 3), dim=1)).item())
        return min(nearest_image_idx // per, self.n_split_dataset - 1), nearest_image_idx

    def split_of_origin(self, origin, direction=None):
        """The eval branch of generate_ray_samples (perssampler.py:365-376): clustered labels when they exist, else the runs."""
        if self.cameras_labels is not None:
            return self.get_nearest_split_dataset(origin, direction)
        return self.get_nearest_split_dataset_orig(origin)

    def query_tree_nodes_centers(self, anchors):
        assert anchors.shape[1] == 1 and anchors.dtype == torch.int64  # perssampler.py:500-501
        return self.sampler.qurey_tree_nodes_centers(anchors)

    def get_distance_matrix_oct(self):
        """perssampler.py:170-214: one ray per training camera along its optical axis; the centres of the nodes it crosses are
        kept (cameras_rays_nodes_centers / cameras_valid_mask, used by get_oct_split_dataset); the matrix the clustering gets
        is the plain camera-position distance matrix (:201-207), built here in one cdist instead of n^2 .item() calls."""
        rays_o = self.c2w[:, 0:3, -1].contiguous()
        rays_d = self.c2w[:, :, 2].contiguous()
        n_cams = rays_o.shape[0]
        dense = self.sampler.GetSamples(rays_o, rays_d, self.bounds)
        anchors = dense[5]
        # the reference indexes QueryTreeNodeCenters with anchors[..., 0] (:189); column 0 is trans_idx, kept as is
        centers = self.query_tree_nodes_centers(anchors[:, :, 0].reshape(-1, 1).contiguous())
        self.cameras_rays_nodes_centers = centers.view(n_cams, self.max_pts_per_ray, 3)
        self.cameras_valid_mask = anchors[:, :, 0] > 0
        distance_matrix = torch.cdist(rays_o.double(), rays_o.double()).float().cpu()
        return distance_matrix, torch.ones(n_cams, dtype=torch.bool)

    def train_cameras_clustering_oct(self, k):
        """perssampler.py:216-242: cameras_labels [n,1] int64, k non-empty equal-size clusters.  The reference's
        SpectralEqualSizeClustering (sklearn, SURVEY 2a OUT) is replaced by gfnerf_amd.cluster's capacity-constrained k-means
        over the same camera positions; the contract (equal sizes, none empty, seed 1234) is the reference's."""
        from .. import cluster

        assert self.cameras_labels is None
        # the reference clusters on get_distance_matrix_oct()'s matrix (:218), which as a side effect fills the buffers
        # cameras_rays_nodes_centers / cameras_valid_mask (:196-197: they travel in its checkpoints and feed get_oct_split_dataset)
        self.get_distance_matrix_oct()
        pos = self.c2w[:, 0:3, -1].detach().cpu().numpy()
        labels = cluster.equal_size_clusters(pos, int(k), seed=1234)
        self.cameras_labels = torch.from_numpy(labels).view(-1, 1).to(torch.int64)
        self.num_cams_per_cluster = [int((self.cameras_labels == i).sum().item()) for i in range(int(k))]
        assert 0 not in self.num_cams_per_cluster

    def update_oct_nodes(self, sampled_anchors, pts_idx_bounds, sampled_weights, sampled_alpha, iter_step: int):
        self.sampler.UpdateOctNodes(sampled_anchors, pts_idx_bounds, sampled_weights, sampled_alpha, iter_step)

    def update_ray_march(self, cur_step: int):
        self.sampler.UpdateRayMarch(cur_step)

    def update_mode(self, mode: int):
        self.sampler.UpdateMode(mode)

    def update_block_idx(self, block_centers):
        self.sampler.UpdateBlockIdxs(block_centers)

    def states(self):
        return self.sampler.States()

    def load_states(self, states, idx):
        return self.sampler.LoadStates(states, idx)

    def state_dict_hook(self, *args):
        destination, prefix = args[1], args[2]
        tree_nodes_gpu_, pers_trans_gpu_, tree_visit_cnt_, milestones_ts_ = self.states()
        destination[prefix + "tree_nodes_gpu"] = tree_nodes_gpu_
        destination[prefix + "pers_trans_gpu"] = pers_trans_gpu_
        destination[prefix + "tree_visit_cnt"] = tree_visit_cnt_
        destination[prefix + "milestones_ts"] = milestones_ts_
        return destination

    def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True):
        """perssampler.py:516-548: consumes persampler.{tree_nodes_gpu,pers_trans_gpu,tree_visit_cnt,milestones_ts} and the
        camera poses the clustering / nearest-split lookups use (field.persampler.c2w there: the sampler is registered a second
        time under the field)."""
        keys = ["persampler." + k for k in ("tree_nodes_gpu", "pers_trans_gpu", "tree_visit_cnt", "milestones_ts")]
        states = [state_dict[k] for k in keys]
        for k in keys:
            del state_dict[k]
        self.load_states(states, idx=0)
        if "field.persampler.c2w" in state_dict:  # :533-534; the copy under persampler.c2w is a registered buffer and loads with the module
            self.c2w = state_dict["field.persampler.c2w"].to(self.device)
            del state_dict["field.persampler.c2w"]
        for k in ("cameras_labels", "cameras_rays_nodes_centers", "cameras_valid_mask"):  # the clustering result, when it was saved
            for pre in ("persampler.", "field.persampler."):
                if pre + k in state_dict:
                    setattr(self, k, state_dict.pop(pre + k))
        return None

    # ------------------------------------------------------------------ the rest of the reference's surface
    def get_oct_split_dataset(self, cam_origin, cam_direction):
        """perssampler.py:289-348 (its call sites are commented out, :377-381): the training camera whose optical-axis ray crosses
        the most similar set of tree nodes -> (split, camera)."""
        assert self.cameras_labels is not None and self.cameras_rays_nodes_centers is not None
        dense = self.sampler.GetSamples(cam_origin.view(1, 3).to(self.device), cam_direction.view(1, 3).to(self.device), self.bounds[0, :].view(1, 2))
        anchors = dense[5]
        cur_centers = self.query_tree_nodes_centers(anchors[:, :, 0].reshape(-1, 1).contiguous())          # [1024,3]
        cur_mask = (anchors[:, :, 0] > 0) & self.cameras_valid_mask                                       # [n_cams,1024]
        dist = torch.norm(cur_centers[None] - self.cameras_rays_nodes_centers, p=2, dim=-1)               # [n_cams,1024]
        cnt = cur_mask.sum(1)
        mean = torch.where(cnt > 0, (dist * cur_mask).sum(1) / cnt.clamp(min=1), torch.full_like(cnt, 1e9, dtype=dist.dtype))
        camera_idx = int(torch.argmin(mean).item())
        return int(self.cameras_labels[camera_idx].item()), camera_idx

    def vis_octree(self, base_exp_dir):
        """perssampler.py:449-455 + PersOctree::VisOctree: camera positions and the wireframe of every leaf with a warp, as .obj."""
        import os

        assert os.path.isdir(base_exp_dir)
        with open(os.path.join(base_exp_dir, "cam_pos.obj"), "w") as f:
            for pos in self.c2w[:, :, 3].cpu().tolist():
                f.write(f"v {pos[0]} {pos[1]} {pos[2]}\n")
        self.sampler.VisOctree(base_exp_dir)  # perssampler.py:455

    def visualize_split_cameras(self, output_dir):
        """perssampler.py:262-283 writes one .ply per camera cluster with plyfile; here the same points as ASCII .ply (positions +
        viewing axes), no third-party writer."""
        import os

        assert self.cameras_labels is not None
        k = int(self.cameras_labels.max().item()) + 1
        for i in range(k):
            m = (self.cameras_labels == i).view(-1).to(self.c2w.device)
            pos, dirs = self.c2w[m, 0:3, -1].cpu().numpy(), self.c2w[m, :, 2].cpu().numpy()
            with open(os.path.join(output_dir, f"camera_cluster_{i}.ply"), "w") as f:
                f.write("ply\nformat ascii 1.0\ncomment vertices\n" + f"element vertex {pos.shape[0]}\n" +
                        "".join(f"property float {c}\n" for c in ("x", "y", "z", "nx", "ny", "nz")) + "end_header\n")
                for p_, d_ in zip(pos, dirs):
                    f.write(" ".join(repr(float(v)) for v in list(p_) + list(d_)) + "\n")

    def get_points_anchors(self, rays_o, rays_d, t_starts, t_ends):
        return self.sampler.get_points_anchors(rays_o, rays_d, t_starts, t_ends)  # refuses: dead proposal-sampler path

    def trans_query_frame(self, world_positions_flat, anchors_flat):
        return self.sampler.trans_query_frame(world_positions_flat, anchors_flat)  # refuses: dead proposal-sampler path

    def get_edge_samples(self, n_pts: int):
        """perssampler.py:495-497: forwards to the native GetEdgeSamples, which refuses (no caller in gfnerf/)."""
        return self.sampler.GetEdgeSamples(n_pts)

    def to_proposal_sampler_oct_tree_nodes(self):
        """perssampler.py:552-581: one record per tree node (the model only takes its length: n_volumes, nerfacto.py:268)."""
        nd = self.sampler._nodes_np()
        tr = self.sampler.pers_trans_gpu_.cpu().numpy().view(_TRANS_DT) if self.sampler.pers_trans_gpu_.numel() else None
        out = []
        for n in nd:
            ti = int(n["trans_idx"])
            trans = None if (ti < 0 or tr is None) else dict(w2xz=torch.tensor(tr["w2xz"][ti]), center=torch.tensor(tr["center"][ti]).unsqueeze(-1),
                                                              weight=torch.tensor(tr["weight"][ti]), dis_summary=float(tr["dis_summary"][ti]))
            out.append(dict(center=torch.tensor(n["center"]).unsqueeze(-1), side_len=float(n["side_len"]), childs=[], parent=-1,
                            is_leaf_node=bool(n["is_leaf_node"]), visi_cams=0, pers_trans=trans))
        return out

    @property
    def sub_div_milestones_(self):
        return self.sampler.get_sub_div_milestones_()

    @property
    def compact_freq_(self):
        return self.sampler.get_compact_freq_()

    @property
    def max_oct_intersect_per_ray_(self):
        return self.sampler.get_max_oct_intersect_per_ray_()

    @property
    def global_near_(self):
        return self.sampler.get_global_near_()

    @property
    def sample_l_(self):
        return self.sampler.get_sample_l_()

    @property
    def scale_by_dis_(self):
        return self.sampler.get_scale_by_dis_()

    @property
    def mode_(self):
        return self.sampler.get_mode_()

    @property
    def n_volumes_(self):
        return self.sampler.get_n_volumes_()

    @property
    def sampled_oct_per_ray_(self):
        return self.sampler.get_sampled_oct_per_ray_()

    @property
    def ray_march_fineness_(self):
        return self.sampler.get_ray_march_fineness_()

    @property
    def tree_nodes_(self):
        nd = self.sampler._nodes_np()
        return TreeNodes(center=torch.tensor(nd["center"]), side_len=torch.tensor(nd["side_len"]).unsqueeze(-1),
                         block_idx=torch.tensor(nd["block_idx"]).unsqueeze(-1), trans_idx=torch.tensor(nd["trans_idx"]).unsqueeze(-1),
                         is_leaf_node=torch.tensor(nd["is_leaf_node"].astype(bool)).unsqueeze(-1))
