"""Host-side inputs of the two render paths: OBJ/MTL meshes, transfer functions, cameras."""
