from .train import run

run()
