"""unreal_amd -- MI355X-native UNREAL/A3C actor-learner hot path (gfx950 HIP kernels behind the
reference's Trainer.process() / UnrealModel / options surface).  See DESIGN.md."""
__version__ = "0.1.0"
