"""materialrefgs_amd -- MI355X-native surfel rasterizer + BRDF shading hot path (drop-in for the reference's
diff_surfel_rasterization / gaussian_renderer.render_* path).  See DESIGN.md."""
__version__ = "0.1.0"
