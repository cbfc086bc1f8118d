"""Model classes mirroring `magicdrive.networks` of the reference (same module and class names,
so hydra configs can point `model.unet_module` / `model.model_module` / `map_embedder_cls` /
`bbox_embedder_cls` at them by dotted path — SURVEY.md §8b B1)."""
