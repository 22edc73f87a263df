"""Drop-in alias: `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:9,
gui/gs_renderer.py:14 of the reference) resolves to the MI355X build."""
