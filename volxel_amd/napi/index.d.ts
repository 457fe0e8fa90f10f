// Typings of the headless host (names follow volxel-3d-viewer/src/viewer.ts, utils/data.ts, common.ts)
export type ColorStop = { color: [number, number, number, number]; stop: number };
export type BrickGridMessage = {            // WasmWorkerMessageDicomReturn, common.ts:37-55
  type: "return_dicom";
  indirectionSize: [number, number, number]; rangeSize: [number, number, number]; atlasSize: [number, number, number];
  transform: Float32Array; histogram: Uint32Array; histogramGradientRange: [number, number];
  histogramGradient: Int32Array; minMaj: [number, number]; indexExtent: [number, number, number];
  rangeMipmaps: { mipmap: Uint16Array; stride: [number, number, number] }[];
  indirection: Uint32Array; range: Uint16Array; atlas: Uint8Array; brickCounter: number;
};
export declare const VolxelRenderMode: { default: 0; no_dda: 1; raymarch: 2; dvr: 3; dvr_phong: 4 };
export declare function generateTransferFunction(colors: ColorStop[], generatedSteps?: number): { data: Float32Array; length: number };
export declare class Camera {
  pos: number[]; view: number[];
  /** [build] null: the reference's perspective camera (scene.ts:65-72); a number: orthographic, image spans +-orthoHalfHeight world units vertically */
  orthoHalfHeight: number | null;
  yaw: number; pitch: number;
  constructor(distance?: number); viewMatrix(): number[]; projMatrix(aspect: number, fov?: number): number[];
  /** scene.ts:15-52 */
  rotateAroundView(by: [number, number]): void; zoom(by: number): boolean;
  translateOnPlane(by: [number, number]): void; translate(by: [number, number, number]): void;
}
export declare class Environment {          // representation/environment.ts; row 0 of `floats` = top
  constructor(floats: Float32Array, width: number, height: number, strength?: number);
  floats: Float32Array; width: number; height: number; strength: number;
  static default(): Environment;
}
export declare class Volxel3DDicomRenderer {
  /** layout: 0 reference textures, 1 cellquad, 2 brickf32, 3 (default) per render mode, 4 bricku8 (8-bit bricks decoded at staging) -- include/volxel_hip.h VxLayout */
  constructor(opts?: { width?: number; height?: number; device?: number; layout?: number; lowResPreview?: boolean });
  environment: Environment | null;
  setEnvironment(env: Environment | null): void;
  settings: Record<string, any>; camera: Camera; envStrength: number; frameIndex: number;
  renderMode: keyof typeof VolxelRenderMode;
  restartFromFiles(files: (string | Uint8Array)[], threads?: number): void;
  /** viewer.ts:977-1040.  ZIP (stored / deflate) and Radiance RGBE are decoded by the host; URLs are paths or file:// URLs
   *  (no fetch in this Node); OpenEXR bytes are refused with a message naming setupEnv(). */
  restartFromZip(zip: string | Uint8Array, threads?: number): void;
  restartFromZipUrl(url: string, threads?: number): void;
  restartFromURLs(urls: string[], threads?: number): void;
  loadEnv(bytes: Uint8Array): void;
  loadEnvFromUrl(url: string): void;
  /** viewer.ts:443-449,543-551,789-795: the light follows the camera when settings.syncLightDir is on */
  maybeSyncLight(): void;
  rotateCamera(by: [number, number]): void;
  syncLightDir: boolean;
  setupEnv(env: { width: number; height: number; floats: Float32Array }): void;
  /** restartFromFiles for slices already read into memory */
  restartFromBytes(files: Uint8Array[], threads?: number): void;
  restartFromVoxels(voxels: Uint16Array, dims: [number, number, number], spacing?: [number, number, number], maxValue?: number, threads?: number): void;
  setupFromGrid(grid: BrickGridMessage): void;
  changeTransferFunc(data: Float32Array, length: number): void;
  restartRendering(): void;
  restoreSettings(settings: any): void;
  /** data-benchmark-url runner (viewer.ts:856-890); returns VolxelBenchmarkResult records */
  startBenchmark(collection: { sharedSettings: any[]; benchmarks: any[] }, volumes?: Record<string, BrickGridMessage>): any[];
  bindUniforms(): { buffer: ArrayBuffer };
  render(frames?: number, inFlight?: number): void;
  probeTileCosts(): Uint32Array;
  setTileOrder(perm: Uint32Array | null): void;
  deviceInfo(): { name: string; computeUnits: number; hbmBytes: number };
  finish(): void;
  readAccum(): Float32Array;
  readDisplay(): Uint8Array;
  counters(): { samples: number; rays: number; pixels: number; skipSteps: number; gradSamples: number; tfSamples: number; activeLaneSlots: number; laneSlots: number; launches: number; frames: number;
                kernelMs: number; lastKernelMs: number; gathers: number; ldsReads: number; mergeMs: number; minLaunchFrames: number; maxLaunchFrames: number };
  resetCounters(): void;
  dispose(): void;
}
/** viewer.ts:1455-1462: keeps the worker factory, returns the element-name -> class table ("volxel-3d-viewer") */
export declare function registerVolxelComponents(worker?: () => unknown): Record<string, typeof Volxel3DDicomRenderer>;
export declare function readZipSlices(zip: Uint8Array): Uint8Array[];
export declare function decodeEnvironment(bytes: Uint8Array): { floats: Float32Array; width: number; height: number };
