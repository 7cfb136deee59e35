"""How this package issues torch.distributed collectives (RCCL over xGMI with the nccl backend, gloo on CPU)."""


def wait_collective(work):
    """Finish a collective that was issued with ``async_op=True``: the caller's stream waits for it (and, on CPU backends, the host).

    Why every collective of this package is issued asynchronously and then waited for, also the ones it needs at once: since PyTorch 2.7 a
    BLOCKING collective of the nccl (= RCCL) backend is launched on the caller's current stream and leaves its completion event there, and the
    backend's watchdog thread keeps polling that event for up to a poll period after the call has returned.  HIP refuses ``hipEventQuery`` on
    an event whose stream is capturing - and the sampler captures its chains on the caller's stream (cmdgen_sample_chain) - so a chain started
    right after a blocking barrier / all-reduce on the same stream made the watchdog throw and the process abort, now and then
    (profiles/r06_o_rccl_watchdog_capture.txt).  Asynchronous collectives run on the backend's own stream, which nothing here ever captures."""
    if work is not None:
        work.wait()
    return work


def host_barrier(group=None, device=None):
    """A barrier the HOST waits for (all ranks have arrived when it returns), issued the way ``wait_collective`` asks for."""
    import torch
    import torch.distributed as dist
    wait_collective(dist.barrier(group=group, async_op=True))
    if dist.get_backend(group) == 'nccl':                 # a device-side barrier: its wait() orders the stream; the host waits for the device
        torch.cuda.synchronize(device)
