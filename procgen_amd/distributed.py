"""One temperature ladder split over the ranks of a torch.distributed group.

Replicas are independent between swap passes, so a ladder of Tl temperatures is cut into
contiguous blocks of replicas, one block per rank (one process per GPU).  The only exchange
step is, once per swap interval, an all-gather of the per-replica energies (8 bytes each;
RCCL over xGMI when the backend is "nccl", gloo on CPUs); every rank then runs the identical
sequential swap pass (reference src/ParallelTempering.Core.fs:168-175) on the gathered
energies with the shared addressed swap draws and updates its replicated slot map.  States
never move: temperature labels do.

A shard is anything with
    run_to_swap_point(max_steps) -> (steps_done, swap_due)
    local_energies(out)           fills a 1-D float64 tensor with this shard's energies, replica order
    swap_pass(all_energies)       1-D float64 tensor, whole ladder, replica order
GraphEngineShard adapts a GraphEngine created with ladder_temp_count / ladder_slot_base.
"""
import torch
import torch.distributed as dist


class GraphEngineShard:
    """A GraphEngine holding replicas [ladder_slot_base, ladder_slot_base + temp_count) of one ladder."""

    def __init__(self, engine):
        assert engine.G == 1, "a split ladder is one ladder"
        self.engine = engine
        self.local_count = engine.T
        self.ladder_count = engine.Tl
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.ExternalStream(engine.stream())        # the engine's own HIP stream, as torch sees it
        self.staged = torch.zeros(self.ladder_count, dtype=torch.float64, device=self.device)
        # the engine's energy buffer (device memory the engine owns for its whole life) as a tensor, without a copy
        ptr, count = engine.local_energy_ptr()
        assert count == self.local_count

        class _Buffer:
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3}
        self._keep = _Buffer()
        self.energy_view = torch.as_tensor(self._keep, device=self.device)

    def run_to_swap_point(self, max_steps):
        return self.engine.run_to_swap_point(max_steps)

    def local_energies(self, out):
        """out: a host or a device tensor.  A device tensor is filled on torch's current stream, which is made to wait
        for the engine's stream (the steps launched so far) on the device: the host does not wait, so the gather that
        follows is queued at once.  A host tensor (gloo) needs the values on the host: that copy waits."""
        assert out.dtype == torch.float64 and out.numel() == self.local_count and out.is_contiguous()
        if out.is_cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
            out.copy_(self.energy_view, non_blocking=True)
        else:
            self.engine.read_energies(out.data_ptr())

    def swap_pass(self, all_energies):
        assert all_energies.dtype == torch.float64 and all_energies.numel() == self.ladder_count and all_energies.is_contiguous()
        if not all_energies.is_cuda:                    # gathered on the host (gloo): stage on the device
            self.staged.copy_(all_energies, non_blocking=False)
            all_energies = self.staged
        # the gather (or the staging copy) ran on torch's stream, the pass runs on the engine's: the engine's
        # stream waits for torch's, the host does not
        self.stream.wait_stream(torch.cuda.current_stream())
        self.engine.swap_pass(all_energies.data_ptr())


class SplitLadder:
    """Drives one shard per rank through `steps` iterations of ParallelTempering.run's loop."""

    def __init__(self, shard, group=None, device=None):
        self.shard = shard
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if shard.ladder_count != shard.local_count * self.world:
            raise ValueError("the ladder must split into equal blocks: ladder_temp_count == temp_count * world_size")
        self.device = device if device is not None else getattr(shard, "device", torch.device("cpu"))
        self.local = torch.zeros(shard.local_count, dtype=torch.float64, device=self.device)
        self.all = torch.zeros(shard.ladder_count, dtype=torch.float64, device=self.device)
        self.gathers = 0

    def run(self, steps):
        left = int(steps)
        while left > 0:
            done, due = self.shard.run_to_swap_point(left)
            left -= done
            if due:
                self.shard.local_energies(self.local)
                if self.world > 1:
                    dist.all_gather_into_tensor(self.all, self.local, group=self.group)
                else:
                    self.all.copy_(self.local)
                self.gathers += 1
                self.shard.swap_pass(self.all)
            elif done == 0:
                raise RuntimeError("the shard made no progress")
