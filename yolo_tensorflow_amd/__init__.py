"""MI355X-native YOLO inference hot path.  `hip` is the ctypes shim over libyolo_hip.so;
`yolo_v3`, `yolo_v2`, `detector` mirror the reference's Python entry points; `darknet_io` handles cfg/.weights;
`dist` shards batches across GPUs."""
__all__ = ["hip", "darknet_io", "yolo_v3", "yolo_v2", "detector", "dist"]
