"""hypelcnn_amd -- MI355X-native hot path of aligokalppeker/hypelcnn behind its plugin API.

Layout mirrors the reference so plugins resolve by the same dotted names
("nnmodel.HYPELCNNModel.HYPELCNNModel", "importer.InMemoryImporter.InMemoryImporter", ...):

  nnmodel/   NNModel plugins (HYPELCNN, DUALCNN, CONCNN) written against hypelcnn_amd.graph
  common/    value objects, create_graph / optimize_nn, metrics, flag groups
  importer/  DataImporter plugins        loader/  DataLoader plugins
  classify/  the training step loop      gan/     shadow GAN stacks and wrappers
  graph.py   deferred symbolic graph     plan.py  lowering to HIP launches
  runtime.py session / optimiser / DP    backend.py  ctypes binding of csrc/libhypel_hip.so
"""
__version__ = "0.1.0"
