"""MI355X-native parallel-tempering engine behind the libgpu_compute C ABI."""
__all__ = ["native", "engine", "workloads"]
